#!/usr/bin/env python3
"""Quick A/B of GEMM variants on a few LLaMA-3-8B shapes (interleaved rounds, median)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
variants = [int(x) for x in os.environ.get("VARIANTS", "6,7").split(",")]
shapes = [("qkv", 16384, 6144, 4096), ("gate_up", 16384, 28672, 4096), ("down", 16384, 4096, 14336), ("dW_gate_up", 28672, 4096, 16384), ("o", 16384, 4096, 4096)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16(); b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ref = None
    res = {v: [] for v in variants}
    for v in variants:
        ops.gemm(a, b, out=c, variant=v)
        if ref is None: ref = c.float().clone()
        else:
            err = float((c.float() - ref).abs().max())
            assert err < 1.0 or v > 90, (name, v, err)
    for rnd in range(5):
        for v in variants:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): ops.gemm(a, b, out=c, variant=v)
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 5 * 1e-3)
    print(name, m, n, k, "  ".join(f"v{v}: {2.0*m*n*k/statistics.median(res[v])/1e12:7.1f} TF" for v in variants), flush=True)
