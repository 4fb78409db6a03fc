#!/usr/bin/env python3
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
M, K = int(os.environ.get("M", 16384)), 4096
for N in (4096, 5120, 6144, 7168, 8192, 12288, 6144, 4096):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(a, b, out=c, variant=11)
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.gemm(a, b, out=c, variant=11)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 5)
    t = statistics.median(ts)
    print(f"M={M} N={N:6d} K={K}: tiles={(M//256)*((N+255)//256):5d} ({(M//256)*((N+255)//256)/256:5.2f} waves)  {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:7.1f} TF/s", flush=True)
