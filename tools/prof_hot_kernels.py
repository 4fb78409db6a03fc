#!/usr/bin/env python3
"""rocprofv3 --pmc target: the hot MFMA kernels of the step at LLaMA-3-8B shapes (3 launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
T = 24576
for name, n, k in [("qkv", 6144, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]:
    x = (torch.randn(T, k, device="cuda") * 0.5).bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
    for _ in range(3): ops.gemm(x, w)
B, L, Hq, Hkv, d = 12, 2048, 32, 8, 128
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
for _ in range(3): o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, True, None)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, True, None, dqkv[:, :Hq * d], dqkv[:, Hq * d:(Hq + Hkv) * d], dqkv[:, (Hq + Hkv) * d:])
torch.cuda.synchronize()
