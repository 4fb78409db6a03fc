#!/usr/bin/env python3
"""mm355_gemm_splitk_bf16 on the decode-wide / prompt-pass shapes (ROWS rows against the LLaMA-3-8B projections), weights rotated so that
nothing stays in L2 / MALL between launches.  With MM355_LIB_PATH=build/splitk_<name>/libmm355.so: the timing builds of tools/build_splitk_tune.sh."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
tot = {}
for M in [int(x) for x in os.environ.get("ROWS", "32,64,128,512").split(",")]:
    row = []
    for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
        a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
        ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(6)]
        for w in ws: ops.gemm_splitk(a, w)
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for w in ws: ops.gemm_splitk(a, w)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / len(ws) * 1e3)
        us = statistics.median(ts)
        tot[M] = tot.get(M, 0.0) + us
        row.append(f"{name} {us:6.1f} us ({N * K * 2 / us / 1e6:4.2f} TB/s, S={int(ops._L().mm355_gemm_splitk_ws_floats(M, N, K)) // (M * N)})")
        del ws
    print(f"M={M:4d}: " + "  ".join(row) + f"   sum {tot[M]:6.1f} us", flush=True)
