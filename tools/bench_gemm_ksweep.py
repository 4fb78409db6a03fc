#!/usr/bin/env python3
"""Fixed cost (prologue + epilogue + dispatch) vs per-K cost of the ping-pong GEMM: time over K at fixed M, N."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
M, N = 16384, 4096          # 1024 tiles = 4 full waves of workgroups
pts = []
for K in (512, 1024, 2048, 4096, 8192, 16384):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(a, b, out=c, variant=11)
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.gemm(a, b, out=c, variant=11)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 5)
    t = statistics.median(ts); pts.append((K, t))
    print(f"K={K:6d}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:7.1f} TF/s", flush=True)
(k1, t1), (k2, t2) = pts[2], pts[-1]
slope = (t2 - t1) / (k2 - k1); icpt = t1 - slope * k1
print(f"per-K slope {slope*1e3*64:.2f} us per 64-wide K tile (4 waves of tiles), fixed {icpt*1e3:.1f} us per launch = {icpt/4*1e3:.1f} us per tile wave; asymptotic {2.0*M*N/slope/1e9:.0f} TF/s")
