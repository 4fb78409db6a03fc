#!/usr/bin/env python3
"""Phase stamps of the attn4 forward (TIMING-ONLY build: tools/build_ablation.sh attn4timing -DMM355_ATTN4_TIMING, run with
MM355_LIB_PATH=build/ablate_attn4timing/libmm355.so): per query block, waves 0 and 3: prologue / head / pipelined loop / tail / idle + sync /
epilogue cycles, and cycles per pipelined tile."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops

B, L, Hq, Hkv, d = int(os.environ.get("B", 16)), int(os.environ.get("L", 2048)), int(os.environ.get("HQ", 32)), 8, 128
CAUSAL = os.environ.get("CAUSAL", "1") == "1"
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
for _ in range(3):
    o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, CAUSAL, None, variant=4)
torch.cuda.synchronize()
raw = lse.cpu().view(B, Hq, L // 256, 256)[..., :32].contiguous().view(torch.int64).view(B, Hq, L // 256, 2, 8).double()
names = ["prologue", "head", "loop", "tail", "idle+sync", "epilogue"]
if os.environ.get("SUMMARY"):
    m = raw[:, :, :, 1].mean(dim=(0, 1, 2))
    tiles = raw[:, :, :, 1, 6].clamp(min=1)
    per = (raw[:, :, :, 1, 2] / tiles).mean()
    print(f"[{os.environ.get('TAG', '')}] B{B} L{L} causal={int(CAUSAL)} wave 3 means: " + " ".join(f"{n}={int(m[i])}" for i, n in enumerate(names)) + f" loop/tile={per:.0f}", flush=True)
    sys.exit(0)
print(f"B{B} L{L} H{Hq}/{Hkv} causal={int(CAUSAL)}: mean cycles over all (sample, head) per query block; wave 0 | wave 3")
for xb in range(L // 256):
    row = []
    for w in range(2):
        m = raw[:, :, xb, w].mean(dim=(0, 1))
        tw = int(m[6])
        per = m[2] / max(tw, 1)
        row.append(" ".join(f"{n}={int(m[i])}" for i, n in enumerate(names)) + f" tw={tw} T={int(m[7])} loop/tile={per:.0f} total={int(m[:6].sum())}")
    print(f"  block {xb}: {row[0]}\n           {row[1]}")
