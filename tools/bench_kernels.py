#!/usr/bin/env python3
"""Micro-benchmarks of the libmm355 kernels on LLaMA-3-8B / SigLIP shapes (needs an MI355X).
Prints one line per (kernel, shape, variant): time and achieved TFLOP/s or GB/s.  Random data."""
import argparse
import json
import sys, os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops

DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=8192)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    M = args.tokens
    res = []
    h, I, V = 4096, 14336, 128258
    shapes = [("qkv", M, 6144, h), ("o", M, h, h), ("gate_up", M, 2 * I, h), ("down", M, h, I),
              ("dW_gate_up", 2 * I, h, M), ("dW_down", h, I, M), ("lm_head", 4096, V, h),
              ("vit_fc1", 729 * 8, 4304, 1152), ("vit_fc2", 729 * 8, 1152, 4304)]
    for name, m, n, k in shapes:
        a = torch.randn(m, k, device=DEV).bfloat16()
        b = torch.randn(n, k, device=DEV).bfloat16()
        ldc = (n + 63) // 64 * 64
        c = torch.empty(m, ldc, device=DEV, dtype=torch.bfloat16)[:, :n]
        for v in range(1, 7):
            if k % 64 and v in (2, 4, 6):
                continue
            try:
                t = timeit(lambda: ops.gemm(a, b, out=c, variant=v))
            except Exception as ex:  # noqa
                print(f"gemm {name} v{v}: ERROR {ex}")
                continue
            tf = 2.0 * m * n * k / t / 1e12
            res.append(dict(kernel="gemm", name=name, M=m, N=n, K=k, variant=v, ms=t * 1e3, tflops=tf))
            print(f"gemm {name:12s} M={m:6d} N={n:6d} K={k:6d} v{v}: {t*1e3:8.3f} ms  {tf:7.1f} TF/s", flush=True)
        del a, b, c
    # attention fwd / bwd, LLaMA-3-8B geometry
    for (B, L) in ((M // 2048, 2048),):
        Hq, Hkv, d = 32, 8, 128
        qkv = (torch.randn(B * L, (Hq + 2 * Hkv) * d, device=DEV) * 0.5).bfloat16()
        q2, k2, v2 = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
        t = timeit(lambda: ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, True, None))
        fl = 4.0 * B * Hq * L * L * d / 2
        print(f"attn_fwd B={B} L={L}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s (causal-halved flops)", flush=True)
        res.append(dict(kernel="attn_fwd", B=B, L=L, ms=t * 1e3, tflops=fl / t / 1e12))
        o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, True, None)
        do = torch.randn_like(o)
        dqkv = torch.empty_like(qkv)
        t = timeit(lambda: ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, True, None,
                                        dqkv[:, :Hq * d], dqkv[:, Hq * d:(Hq + Hkv) * d], dqkv[:, (Hq + Hkv) * d:]), iters=5)
        print(f"attn_bwd B={B} L={L}: {t*1e3:8.3f} ms {2.5*fl/t/1e12:7.1f} TF/s (incl. prep+transposes)", flush=True)
        res.append(dict(kernel="attn_bwd", B=B, L=L, ms=t * 1e3, tflops=2.5 * fl / t / 1e12))
    # HBM-bound kernels
    x = torch.randn(M, h, device=DEV).bfloat16()
    w = torch.ones(h, device=DEV, dtype=torch.bfloat16)
    t = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-5))
    print(f"rmsnorm_fwd M={M}: {t*1e6:8.1f} us {2*M*h*2/t/1e9:8.1f} GB/s", flush=True)
    res.append(dict(kernel="rmsnorm_fwd", M=M, us=t * 1e6, gbps=2 * M * h * 2 / t / 1e9))
    dw = torch.zeros(h, device=DEV)
    t = timeit(lambda: ops.rmsnorm_bwd(x, x, w, 1e-5, dres=x, dw_f32=dw))
    print(f"rmsnorm_bwd M={M}: {t*1e6:8.1f} us {4*M*h*2/t/1e9:8.1f} GB/s", flush=True)
    res.append(dict(kernel="rmsnorm_bwd", M=M, us=t * 1e6, gbps=4 * M * h * 2 / t / 1e9))
    gu = torch.randn(M, 2 * I, device=DEV).bfloat16()
    t = timeit(lambda: ops.swiglu_fwd(gu, I))
    print(f"swiglu_fwd  M={M}: {t*1e6:8.1f} us {3*M*I*2/t/1e9:8.1f} GB/s", flush=True)
    res.append(dict(kernel="swiglu_fwd", M=M, us=t * 1e6, gbps=3 * M * I * 2 / t / 1e9))
    t = timeit(lambda: ops.transpose(gu))
    print(f"transpose   {M}x{2*I}: {t*1e6:8.1f} us {2*M*2*I*2/t/1e9:8.1f} GB/s", flush=True)
    res.append(dict(kernel="transpose", M=M, us=t * 1e6, gbps=2 * M * 2 * I * 2 / t / 1e9))
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
