#!/usr/bin/env python3
"""Variant 12 (two-phase ping-pong, 32-MFMA clusters) against variant 11 (four phases of 16): bit-equality on awkward shapes, then
TFLOP/s on the GEMM shapes of a LLaMA-3-8B decoder layer at 32 768 tokens (interleaved rounds, median)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
for (m, n, k) in [(256, 256, 128), (256, 512, 256), (300, 520, 384), (4096, 4096, 128), (4096, 4104, 4096), (33, 8192, 1024), (8192, 6144, 4096)]:
    a = (torch.randn(m, k, device="cuda", generator=g) * 0.5).bfloat16()
    b = (torch.randn(n, k, device="cuda", generator=g) * 0.5).bfloat16()
    ldc = (n + 7) // 8 * 8
    c11 = torch.full((m, ldc), 7.0, device="cuda", dtype=torch.bfloat16)[:, :n]
    c12 = torch.full((m, ldc), 7.0, device="cuda", dtype=torch.bfloat16)[:, :n]
    ops.gemm(a, b, out=c11, variant=11)
    ops.gemm(a, b, out=c12, variant=12)
    torch.cuda.synchronize()
    ref = a.float() @ b.float().t()
    e11 = float((c11.float() - ref).abs().max()); e12 = float((c12.float() - ref).abs().max())
    print(f"{m}x{n}x{k}: bit-equal {bool(torch.equal(c11, c12))}  max|err| v11 {e11:.3f} v12 {e12:.3f} (|ref| max {float(ref.abs().max()):.1f})", flush=True)
T = 32768
shapes = [("qkv", T, 6144, 4096), ("o", T, 4096, 4096), ("gate_up", T, 28672, 4096), ("down", T, 4096, 14336),
          ("dX_down", T, 14336, 4096), ("dX_gate_up", T, 4096, 28672), ("dW_gate_up", 28672, 4096, T), ("dW_down", 4096, 14336, T)]
tot = {11: 0.0, 12: 0.0}; fl = 0.0
for name, m, n, k in shapes:
    a = (torch.randn(m, k, device="cuda", generator=g) * 0.5).bfloat16()
    b = (torch.randn(n, k, device="cuda", generator=g) * 0.5).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = {11: [], 12: []}
    for v in (11, 12): ops.gemm(a, b, out=c, variant=v)
    for _ in range(5):
        for v in (11, 12):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4): ops.gemm(a, b, out=c, variant=v)
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 4 * 1e-3)
    t11, t12 = statistics.median(res[11]), statistics.median(res[12])
    tot[11] += t11; tot[12] += t12; fl += 2.0 * m * n * k
    print(f"{name:11s}: v11 {2.0*m*n*k/t11/1e12:7.1f} TF   v12 {2.0*m*n*k/t12/1e12:7.1f} TF   ({t11/t12:.3f}x)", flush=True)
    del a, b, c
print(f"layer: v11 {fl/tot[11]/1e12:.1f} TF  v12 {fl/tot[12]/1e12:.1f} TF  ({tot[11]/tot[12]:.3f}x)")
