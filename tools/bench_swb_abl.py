#!/usr/bin/env python3
"""Duration of mm355_gemm_swiglu_bwd_bf16 at the bench shape (32 768 tokens, K = 4096, I = 14336) under the library MM355_LIB_PATH selects
(the product build, or a timing-only build of tools/build_swb_abl.sh).  One line per run; tools/gpu_round6_profiles.sh swbabl runs all four."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
T, K, I = int(os.environ.get("TOKENS", 32768)), 4096, 14336
g = torch.Generator(device="cuda").manual_seed(1)
dy = (torch.randn(T, K, device="cuda", generator=g) * 0.5).bfloat16()
w = (torch.randn(I, K, device="cuda", generator=g) * 0.03).bfloat16()
gu = (torch.randn(T, 2 * I, device="cuda", generator=g) * 1.5).bfloat16()
for _ in range(3):
    ops.gemm_swiglu_bwd(dy, w, gu, I)
ts = []
for _ in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.gemm_swiglu_bwd(dy, w, gu, I)
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 5)
ms = statistics.median(ts)
print(f"{os.environ.get('MM355_LIB_PATH', 'product build'):34s} fused launch {ms:.3f} ms  ({2.0 * T * K * I / ms / 1e9:.0f} TFLOP/s of GEMM flops; epilogue traffic {T * I * 2 * 3.5 / 1e9:.2f} GB)")
