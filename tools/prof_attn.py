#!/usr/bin/env python3
"""Runs only the attention kernels a few times (target for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
B, L, Hq, Hkv, d = 4, 2048, 32, 8, 128
causal = os.environ.get("CAUSAL", "1") == "1"
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
for _ in range(3):
    o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, causal, None)
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, causal, None, dqkv[:, :Hq * d], dqkv[:, Hq * d:(Hq + Hkv) * d], dqkv[:, (Hq + Hkv) * d:])
torch.cuda.synchronize()
