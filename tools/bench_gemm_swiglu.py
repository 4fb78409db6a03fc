#!/usr/bin/env python3
"""gate|up GEMM + SwiGLU: two launches vs the fused launch (LLaMA-3-8B, 32 768 tokens), interleaved rounds, median."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
T, K, I = int(os.environ.get("TOKENS", 32768)), 4096, 14336
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.randn(T, K, device="cuda", generator=g) * 0.5).bfloat16()
w = (torch.randn(2 * I, K, device="cuda", generator=g) * 0.03).bfloat16()
def two():
    gu = ops.gemm(x, w); return gu, ops.swiglu_fwd(gu, I)
def one():
    return ops.gemm_swiglu(x, w, I)
a, b = two(), one()
print("bit-equal gu", bool(torch.equal(a[0], b[0])), "act", bool(torch.equal(a[1], b[1])))
res = {"two": [], "fused": [], "gemm_only": []}
for _ in range(6):
    for name, fn in (("two", two), ("fused", one), ("gemm_only", lambda: ops.gemm(x, w))):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): fn()
        e.record(); torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 3)
print("  ".join(f"{k}: {statistics.median(v):.3f} ms" for k, v in res.items()))
