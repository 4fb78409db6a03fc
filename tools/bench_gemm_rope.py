#!/usr/bin/env python3
"""q|k|v GEMM + RoPE: two launches vs the fused launch (LLaMA-3-8B, 16 x 2048 tokens)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
B, L, Hq, Hkv, d, K = 16, 2048, 32, 8, 128, 4096
N = (Hq + 2 * Hkv) * d
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.randn(B * L, K, device="cuda", generator=g) * 0.5).bfloat16()
w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).bfloat16()
cos, sin = ops.rope_table(L, d, 500000.0, "cuda")
def two():
    q = ops.gemm(x, w); ops.rope_qk_(q, B, L, Hq, Hkv, d, cos, sin); return q
def one():
    return ops.gemm_rope(x, w, B, L, Hq, Hkv, d, cos, sin)
print("bit-equal", bool(torch.equal(two(), one())))
res = {"two": [], "fused": [], "gemm_only": []}
for _ in range(6):
    for name, fn in (("two", two), ("fused", one), ("gemm_only", lambda: ops.gemm(x, w))):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4): fn()
        e.record(); torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 4)
print("  ".join(f"{k}: {statistics.median(v):.3f} ms" for k, v in res.items()))
