#!/usr/bin/env python3
"""down_proj input-gradient GEMM + SwiGLU backward: two launches vs the fused launch (LLaMA-3-8B, 32 768 tokens)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
T, K, I = int(os.environ.get("TOKENS", 32768)), 4096, 14336
g = torch.Generator(device="cuda").manual_seed(1)
dy = (torch.randn(T, K, device="cuda", generator=g) * 0.5).bfloat16()
w = (torch.randn(I, K, device="cuda", generator=g) * 0.03).bfloat16()
gu = (torch.randn(T, 2 * I, device="cuda", generator=g) * 1.5).bfloat16()
def two():
    return ops.swiglu_bwd_t(gu, ops.gemm(dy, w), I)
def one():
    return ops.gemm_swiglu_bwd(dy, w, gu, I)
a, b = two(), one()
print("bit-equal dgu", bool(torch.equal(a[0], b[0])), "actT", bool(torch.equal(a[1], b[1])), "dguT", bool(torch.equal(a[2], b[2])))
if not torch.equal(a[0], b[0]):
    d = (a[0].float() - b[0].float()).abs(); print("  dgu differing fraction", float((d > 0).float().mean()), "max", float(d.max()))
del a, b
res = {"two": [], "fused": [], "gemm_only": []}
for _ in range(6):
    for name, fn in (("two", two), ("fused", one), ("gemm_only", lambda: ops.gemm(dy, w))):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): fn()
        e.record(); torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 3)
print("  ".join(f"{k}: {statistics.median(v):.3f} ms" for k, v in res.items()))
