#!/usr/bin/env python3
"""mm355_attn_decode by cache length, batch and kernel variant (0 = product: 1024-thread workgroups, one per query head -- pairs from 64
workgroups on -- while the bound on the cached lengths is <= 1024 rows, one per GQA group beyond; 1 = 256-row chunks; 2 = one per GQA
group whatever the bound; VARS=0,1,2): launches for rocprofv3
(`rocprofv3 --kernel-trace -d DIR -o ad -- python tools/bench_attn_decode.py`, then tools/rocpd_kernels.py DIR/ad_results.db --runs): the
wall times printed here are launch-bound (~19 us per Python call), only the profiler's kernel durations mean anything."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops

DEV = "cuda"
Hq, Hkv, d = 32, 8, 128


def t(fn, it=int(os.environ.get("IT", 50))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for B in [int(b) for b in os.environ.get("BS", "1,8").split(",")]:
    for kv in [int(v) for v in os.environ.get("KVS", "100,256,500,600,1000,1100,2000,4000").split(",")]:
        cap = kv + 8
        q = (torch.randn(B, Hq * d, device=DEV) * 0.5).bfloat16()
        kc = (torch.randn(B, cap, Hkv * d, device=DEV) * 0.5).bfloat16()
        vc = (torch.randn(B, cap, Hkv * d, device=DEV) * 0.5).bfloat16()
        lens = torch.full((B,), kv, dtype=torch.int32, device=DEV)
        ws = torch.zeros(int(ops._L().mm355_attn_decode_ws_floats(B, Hq, d, cap)), device=DEV, dtype=torch.float32)
        out = torch.empty(B, Hq * d, device=DEV, dtype=torch.bfloat16)
        row = []
        ref = ops.attn_decode(q, kc, vc, lens, cap, Hq, Hkv, d, d ** -0.5, workspace=ws, variant=0).float()
        for var in [int(v) for v in os.environ.get("VARS", "0,1").split(",")]:
            us = t(lambda: ops.attn_decode(q, kc, vc, lens, cap, Hq, Hkv, d, d ** -0.5, out=out, workspace=ws, variant=var))
            row.append(f"v{var} {us:6.1f} (|d| {float((out.float() - ref).abs().max()):.1e})")
        print(f"B={B} kv={kv:5d}: " + "  ".join(row) + " us", flush=True)
