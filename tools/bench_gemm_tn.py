#!/usr/bin/env python3
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops
for name, m, n, k in [("dW_gate_up", 28672, 4096, 16384), ("dW_down", 4096, 14336, 16384), ("dW_qkv", 6144, 4096, 16384), ("dW_o", 4096, 4096, 16384)]:
    at = torch.randn(k, m, device="cuda").bfloat16(); bt = torch.randn(k, n, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    a2, b2 = at.t().contiguous(), bt.t().contiguous()
    def tn(): ops.gemm_tn(at, bt, c)
    def nt(): ops.gemm(a2, b2, out=c)
    def nt_tr(): ops.gemm(ops.transpose(at), ops.transpose(bt), out=c)
    res = {}
    for nm, fn in (("tn", tn), ("nt(pre-transposed)", nt), ("transpose+nt", nt_tr)):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3): fn()
            e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 3 * 1e-3)
        res[nm] = 2.0 * m * n * k / statistics.median(ts) / 1e12
    print(name, m, n, k, "  ".join(f"{k_}: {v:7.1f} TF" for k_, v in res.items()), flush=True)
