#!/usr/bin/env python3
"""rocprofv3 target: one gate_up-sized launch of each gradient GEMM form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
T, n_out, k_in = 16384, 8192, 4096
dy = (torch.randn(T, n_out, device="cuda") * 0.5).bfloat16()
x = (torch.randn(T, k_in, device="cuda") * 0.5).bfloat16()
w = (torch.randn(n_out, k_in, device="cuda") * 0.02).bfloat16()
dw = torch.empty(n_out, k_in, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm_tn(dy, x, dw)          # TN  <true,true>
    ops.gemm_nn(dy, w)              # NN  <false,true>
    ops.gemm(x, w)                  # NT  <false,false>
torch.cuda.synchronize()
