#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
def t(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it
B, L, Hq, Hkv, d = int(os.environ.get('B', 4)), int(os.environ.get('L', 2048)), int(os.environ.get('HQ', 32)), 8, 128
CAUSAL = os.environ.get('CAUSAL', '1') == '1'
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
fl = 4.0 * B * Hq * L * L * d / (2 if CAUSAL else 1)
ms = t(lambda: ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, CAUSAL, None))
print(f"[{os.environ.get('TAG','')}] attn_fwd {ms:.3f} ms {fl/ms/1e9:.1f} TF/s", flush=True)
o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, CAUSAL, None)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
ms = t(lambda: ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, CAUSAL, None, dqkv[:, :Hq * d], dqkv[:, Hq * d:(Hq + Hkv) * d], dqkv[:, (Hq + Hkv) * d:]), 5)
print(f"[{os.environ.get('TAG','')}] attn_bwd(all) {ms:.3f} ms {2.5*fl/ms/1e9:.1f} TF/s", flush=True)
