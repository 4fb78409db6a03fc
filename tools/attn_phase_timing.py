#!/usr/bin/env python3
"""Reads the phase timestamps a -DMM355_ATTN_TIMING build of attn3::fwd_kernel leaves in the lse rows (timing-only build:
MM355_LIB_PATH=build/ablate_timing/libmm355.so).  Prints cycles per phase by query block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
B, L, Hq, Hkv, d = int(os.environ.get('B', 12)), 2048, 32, 8, 128
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
for _ in range(3):
    o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, True, None)
torch.cuda.synchronize()
w = lse.view(B * Hq, L // 128, 128)[:, :, :16].contiguous().view(torch.int64).view(B * Hq, L // 128, 8).cpu().double()
print("x  ntiles  prologue  loop(ntiles-2)  per-tile  diag2  per-diag-tile  epilogue   total")
for x in range(L // 128):
    r = w[:, x].mean(0)
    nt = int(r[4])
    per = r[1] / max(nt - 2, 1)
    print(f"{x:2d} {nt:5d} {r[0]:9.0f} {r[1]:12.0f} {per:10.0f} {r[2]:8.0f} {r[2]/2:10.0f} {r[3]:10.0f} {float(r[0]+r[1]+r[2]+r[3]):9.0f}")
tot = w[:, :, :4].sum(-1)
print("sum over all blocks (cycles, wave 0 of each):", float(tot.sum()), " prologue share", float(w[:, :, 0].sum() / tot.sum()),
      " epilogue share", float(w[:, :, 3].sum() / tot.sum()), " diag share", float(w[:, :, 2].sum() / tot.sum()))
span = (w[:, :, 7].max() - w[:, :, 6].min())
print("kernel span (cycles):", float(span), " mean busy per slot:", float(tot.sum()) / 512)
