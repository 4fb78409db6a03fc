#!/usr/bin/env python3
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
for M in [int(m) for m in os.environ.get("ROWS", "1,2,4,8,16").split(",")]:
    for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 128258, 4096)]:
        ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(4)]      # rotate: defeat the 256-MiB Infinity Cache
        x = (torch.randn(M, K, device="cuda")).bfloat16()
        i = [0]
        def f():
            ops.gemv(x, ws[i[0] % 4]); i[0] += 1
        ms = t(f)
        print(f"gemv M={M:2d} {name:8s} N={N:6d} K={K:6d}: {ms*1e3:7.1f} us  {N*K*2/ms/1e9:7.2f} TB/s", flush=True)
        del ws
