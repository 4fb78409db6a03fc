#!/usr/bin/env python3
"""Phase stamps of the attn4 dK/dV kernel (TIMING-ONLY build, tools/build_attn4_abl.sh; MM355_LIB_PATH=build/ablate_a4_<name>/libmm355.so):
wave 0 of every workgroup: prologue / head / pipelined loop / tail + sync / epilogue cycles and cycles per pipelined step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
B, L, Hq, Hkv, d = int(os.environ.get("B", 16)), int(os.environ.get("L", 2048)), 32, 8, 128
CAUSAL = os.environ.get("CAUSAL", "1") == "1"
nq, nk = Hq * d, Hkv * d
qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device="cuda") * 0.5).bfloat16()
q2, k2, v2 = qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:]
o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, CAUSAL, None, variant=3)
do = torch.randn_like(o)
dqkv = torch.zeros_like(qkv)
for _ in range(3):
    ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, CAUSAL, None, dqkv[:, :nq], dqkv[:, nq:nq + nk], dqkv[:, nq + nk:], variant=4)
torch.cuda.synchronize()
dk = dqkv[:, nq:nq + nk].contiguous().view(B, L, Hkv, d)[:, ::128].contiguous()        # first row of every 128-key block
raw = dk.cpu().view(B, L // 128, Hkv, d)[..., :24].contiguous().view(torch.int64).view(B, L // 128, Hkv, 6).double()
names = ["prologue", "head", "loop", "tail+sync", "epilogue"]
tag = os.environ.get("TAG", "")
for xb in ([0, L // 256, L // 128 - 1] if not os.environ.get("SUMMARY") else []):
    m = raw[:, xb].mean(dim=(0, 1))
    print(f"[{tag}] key block {xb}: " + " ".join(f"{n}={int(m[i])}" for i, n in enumerate(names)) + f" steps={int(m[5])} loop/step={m[2] / max(m[5] - 1, 1):.0f}")
m = raw.mean(dim=(0, 1, 2))
per = (raw[..., 2] / (raw[..., 5] - 1).clamp(min=1)).mean()
print(f"[{tag}] B{B} L{L} causal={int(CAUSAL)} dK/dV means: " + " ".join(f"{n}={int(m[i])}" for i, n in enumerate(names)) + f" steps={m[5]:.1f} loop/step={per:.0f}", flush=True)
