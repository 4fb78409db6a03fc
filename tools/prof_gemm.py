#!/usr/bin/env python3
"""Runs one GEMM shape a few times per variant (target for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
M, N, K = [int(x) for x in os.environ.get("MNK", "8192,28672,4096").split(",")]
variants = [int(x) for x in os.environ.get("VARIANTS", "6").split(",")]
a = torch.randn(M, K, device="cuda").bfloat16()
b = torch.randn(N, K, device="cuda").bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for v in variants:
    for _ in range(3):
        ops.gemm(a, b, out=c, variant=v)
torch.cuda.synchronize()
