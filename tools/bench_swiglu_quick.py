#!/usr/bin/env python3
"""swiglu forward / backward(+transposed copies) streaming rate at the bench shape (A/B with MM355_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
M, I = 32768, 14336
torch.manual_seed(0)
gu = (torch.randn(M, 2 * I, device="cuda") * 0.8).bfloat16()
da = (torch.randn(M, I, device="cuda") * 0.8).bfloat16()
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
tag = os.environ.get("TAG", "")
ms = t(lambda: ops.swiglu_fwd(gu, I))
print(f"[{tag}] swiglu_fwd   {ms*1e3:7.1f} us  {3.0*M*I*2/ms/1e9:.2f} TB/s", flush=True)
ms = t(lambda: ops.swiglu_bwd_t(gu, da, I))
print(f"[{tag}] swiglu_bwd_t {ms*1e3:7.1f} us  {8.0*M*I*2/ms/1e9:.2f} TB/s", flush=True)
a = ops.swiglu_fwd(gu, I); d, aT, dT = ops.swiglu_bwd_t(gu, da, I)
print(f"[{tag}] checksums {float(a.float().abs().sum()):.6e} {float(d.float().abs().sum()):.6e}")
