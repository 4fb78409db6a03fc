#!/usr/bin/env python3
"""rocprofv3 target: attention forward / backward at the bench shape, variants 3 (attn3) and 4 (attn4 streams), a few launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
B, L, Hq, Hkv, d = int(os.environ.get("B", 16)), int(os.environ.get("L", 2048)), 32, 8, 128
nq, nk = Hq * d, Hkv * d
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:]
dqkv = torch.empty_like(qkv)
for var in (3, 4):
    for _ in range(4):
        o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, True, None, variant=var)
    do = torch.randn_like(o)
    for _ in range(4):
        ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, True, None, dqkv[:, :nq], dqkv[:, nq:nq + nk], dqkv[:, nq + nk:], variant=var)
torch.cuda.synchronize()
