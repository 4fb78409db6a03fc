#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
B, L, Hq, Hkv, d = 8, 2048, 32, 8, 128
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, True, None)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
f = lambda: ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, True, None, dqkv[:, :Hq * d], dqkv[:, Hq * d:(Hq + Hkv) * d], dqkv[:, (Hq + Hkv) * d:])
for _ in range(3): f()
torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
print(f"ABL={os.environ.get('MM355_ATTN_ABL','0')}: attn_bwd(all) {s.elapsed_time(e)/10:.3f} ms", flush=True)
