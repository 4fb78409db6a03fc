#!/usr/bin/env python3
"""One launch each of the three big GEMM shapes of a LLaMA-3-8B layer at 32 768 tokens (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
T = 32768
v = int(os.environ.get("VARIANT", "11"))
for name, m, n, k in [("gate_up", T, 28672, 4096), ("down", T, 4096, 14336), ("dW_gate_up", 28672, 4096, T)]:
    a = (torch.randn(m, k, device="cuda", generator=g) * 0.5).bfloat16()
    b = (torch.randn(n, k, device="cuda", generator=g) * 0.5).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, out=c, variant=v)
    torch.cuda.synchronize()
    del a, b, c
