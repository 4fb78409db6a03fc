#!/usr/bin/env python3
"""Yardstick ONLY (never used by the product): the LLaMA-3-8B GEMM shapes of one decoder layer through the vendor library that
PyTorch-ROCm dispatches bf16 matmuls to (hipBLASLt / rocBLAS), next to libmm355's mm355_gemm_bf16, same box, same random data.
usage: python tools/bench_gemm_vendor.py [tokens]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
h, I, nqkv = 4096, 14336, 6144
shapes = [("qkv fwd", M, nqkv, h), ("o_proj fwd", M, h, h), ("gate_up fwd", M, 2 * I, h), ("down fwd", M, h, I),
          ("d act (dX down)", M, I, h), ("d n2 (dX gate_up)", M, h, 2 * I), ("d n1 (dX qkv)", M, h, nqkv),
          ("dW gate_up", 2 * I, h, M), ("dW down", h, I, M), ("dW qkv", nqkv, h, M), ("dW o", h, h, M)]


def t(fn, it=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


print(f"tokens M = {M}; uniform random bf16 operands; TFLOP/s (ms)")
print(f"{'shape':22s} {'M x N x K':>24s} {'mm355':>18s} {'torch.matmul (vendor)':>24s}  ratio")
tot = [0.0, 0.0, 0.0]
for name, m, n, k in shapes:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * m * n * k
    t1 = t(lambda: ops.gemm(a, b, out=out))
    t2 = t(lambda: torch.matmul(a, b.t(), out=out))
    tot[0] += fl; tot[1] += t1; tot[2] += t2
    print(f"{name:22s} {m:7d} x {n:6d} x {k:6d} {fl / t1 / 1e9:9.1f} ({t1:6.3f}) {fl / t2 / 1e9:14.1f} ({t2:6.3f})  {t2 / t1:5.2f}")
    del a, b, out
print(f"{'layer total':22s} {'':24s} {tot[0] / tot[1] / 1e9:9.1f} ({tot[1]:6.3f}) {tot[0] / tot[2] / 1e9:14.1f} ({tot[2]:6.3f})  {tot[2] / tot[1]:5.2f}")
