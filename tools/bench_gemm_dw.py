#!/usr/bin/env python3
"""A/B of the gradient GEMM forms on LLaMA-3-8B shapes: transposes + NT ping-pong vs contraction-major ping-pong."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops, functional as F

def t(fn, it=6):
    for _ in range(2): fn()
    ts = []
    for _ in range(it):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)

T = int(os.environ.get("TOKENS", 16384))
for name, n_out, k_in in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]:
    dy = (torch.randn(T, n_out, device="cuda") * 0.5).bfloat16()
    x = (torch.randn(T, k_in, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(n_out, k_in, device="cuda") * 0.02).bfloat16()
    dw = torch.empty(n_out, k_in, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * T * n_out * k_in
    a = t(lambda: ops.gemm(F.transpose_padded(dy), F.transpose_padded(x), out=dw))
    b = t(lambda: ops.gemm_tn(dy, x, dw))
    c = t(lambda: ops.gemm(dy, F.transpose_padded(w)))
    d = t(lambda: ops.gemm_nn(dy, w))
    print(f"{name:8s} dW: transposes+NT {a:7.3f} ms ({fl/a/1e9:6.0f} TF)  TN {b:7.3f} ms ({fl/b/1e9:6.0f} TF) | dX: transpose+NT {c:7.3f} ms ({fl/c/1e9:6.0f} TF)  NN {d:7.3f} ms ({fl/d/1e9:6.0f} TF)", flush=True)
