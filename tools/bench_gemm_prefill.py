#!/usr/bin/env python3
"""Prompt-pass GEMM shapes (M = 128 / 512 / 1024 rows against the LLaMA-3-8B projections): every tiling the library has, same box."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
from metamorph_amd.lib import Mm355Error
for M in [int(x) for x in os.environ.get("ROWS", "128,512,1024").split(",")]:
    for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
        a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
        ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(4)]      # rotate weights: no L2 / MALL residency between launches
        out = []
        for v in (0, 2, 9, 7, 11):
            try:
                for w in ws: ops.gemm(a, w, variant=v)
                ts = []
                for _ in range(5):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for w in ws: ops.gemm(a, w, variant=v)
                    e.record(); torch.cuda.synchronize()
                    ts.append(s.elapsed_time(e) / len(ws) * 1e3)
                us = statistics.median(ts)
                out.append(f"v{v}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:6.0f} TF/s, W at {N * K * 2 / us / 1e6:4.2f} TB/s)")
            except Mm355Error:
                out.append(f"v{v}: unsupported")
        print(f"M={M:5d} {name:8s} " + "  ".join(out), flush=True)
