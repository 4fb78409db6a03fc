#!/usr/bin/env python3
"""A/B of one ping-pong GEMM process-level knob (env, read once per process: MM355_GEMM_GM, the raster group height; the early hand-over
knob MM355_GEMM_TAIL this script was written for lost 7 % and was removed again, profiles/r3_gemm_ablation_and_pmc.md) on the GEMM shapes
of a LLaMA-3-8B decoder layer at 32 768 tokens: TFLOP/s per shape (median of interleaved rounds) + a checksum of every output (a knob must
not change a single bit).  Run once per setting:  MM355_GEMM_GM=8 python tools/bench_gemm_tail.py"""
import os, sys, statistics, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
T = int(os.environ.get("TOKENS", "32768"))
shapes = [("qkv", T, 6144, 4096), ("o", T, 4096, 4096), ("gate_up", T, 28672, 4096), ("down", T, 4096, 14336),
          ("dX_down", T, 14336, 4096), ("dX_gate_up", T, 4096, 28672), ("dW_gate_up", 28672, 4096, T), ("dW_down", 4096, 14336, T)]
g = torch.Generator(device="cuda").manual_seed(1)
tot_f, tot_t, sums = 0.0, 0.0, []
for name, m, n, k in shapes:
    a = (torch.randn(m, k, device="cuda", generator=g) * 0.5).bfloat16()
    b = (torch.randn(n, k, device="cuda", generator=g) * 0.5).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, b, out=c)
    sums.append(int(c.view(torch.int16).to(torch.int64).sum()))
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            ops.gemm(a, b, out=c)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 4 * 1e-3)
    t = statistics.median(ts)
    tot_f += 2.0 * m * n * k; tot_t += t
    print(f"{name:11s} {m:6d} x {n:6d} x {k:6d}: {t * 1e3:7.3f} ms {2.0 * m * n * k / t / 1e12:7.1f} TF", flush=True)
    del a, b, c
print(f"TAIL={os.environ.get('MM355_GEMM_TAIL', 'default')} GM={os.environ.get('MM355_GEMM_GM', 'default')}: layer {tot_t * 1e3:.3f} ms "
      f"{tot_f / tot_t / 1e12:.1f} TF  checksum {zlib.crc32(repr(sums).encode()):08x}")
