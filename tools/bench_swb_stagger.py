#!/usr/bin/env python3
"""mm355_gemm_swiglu_bwd_bf16 at the bench shape (A/B with MM355_LIB_PATH: start-stagger builds), and the same kernel on 56 tiles only
(one per CU on 56 CUs: the epilogue without 255 other CUs storing at the same moment) against the plain GEMM of that shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
I, h = 14336, 4096
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
tag = os.environ.get("TAG", "")
for M in (32768, 256, 1024):
    torch.manual_seed(0)
    dy = (torch.randn(M, h, device="cuda") * 0.5).bfloat16()
    wdT = (torch.randn(I, h, device="cuda") * 0.05).bfloat16()
    gu = (torch.randn(M, 2 * I, device="cuda") * 0.8).bfloat16()
    ms = t(lambda: ops.gemm_swiglu_bwd(dy, wdT, gu, I))
    ms0 = t(lambda: ops.gemm(dy, wdT, variant=11))
    dgu, aT, dT = ops.gemm_swiglu_bwd(dy, wdT, gu, I)
    print(f"[{tag}] M={M}: fused swiglu_bwd {ms*1e3:8.1f} us   plain GEMM of the shape {ms0*1e3:8.1f} us   checksum {float(dgu.float().abs().sum()):.6e} {float(dT.float().abs().sum()):.6e}", flush=True)
