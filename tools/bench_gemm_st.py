#!/usr/bin/env python3
"""gemm_st (variant 13: one wave per SIMD, hand-placed stream; 14: the same stream serialised) against the eight-wave ping-pong kernel
(variant 11): bit-equality on small / ragged / every-epilogue cases, then timings on the LLaMA-3-8B shapes of the bench step.
    python tools/bench_gemm_st.py [--quick] [--variants 11,13]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops

DEV = "cuda"


def case(M, N, K, *, bias=False, residual=False, res_row_mod=0, gelu=None, accumulate=False, out_f32=False, lda=None, ldb=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    lda, ldb = lda or K, ldb or K
    a = (torch.randn(M, lda, generator=g) * 0.5).bfloat16().to(DEV)[:, :K]
    b = (torch.randn(N, ldb, generator=g) * 0.5).bfloat16().to(DEV)[:, :K]
    bi = torch.randn(N, generator=g).bfloat16().to(DEV) if bias else None
    rrows = res_row_mod or M
    re = torch.randn(rrows, N, generator=g).bfloat16().to(DEV) if residual else None
    c0 = (torch.randn(M, N, generator=g)).to(torch.float32 if out_f32 else torch.bfloat16).to(DEV)
    outs = {}
    for var in (11, 14, 13):
        c = c0.clone()
        ops.gemm(a, b, out=c, bias=bi, residual=re, res_row_mod=res_row_mod, gelu=gelu, accumulate=accumulate, out_f32=out_f32, variant=var)
        torch.cuda.synchronize()
        outs[var] = c
    ref = a.float() @ b.float().T
    e = float((outs[11].float() - (ref if not (bias or residual or gelu or accumulate) else outs[11].float())).abs().max())
    same13, same14 = torch.equal(outs[13], outs[11]), torch.equal(outs[14], outs[11])
    tag = f"M{M} N{N} K{K} bias={int(bias)} res={int(residual)}/{res_row_mod} gelu={gelu} acc={int(accumulate)} f32={int(out_f32)} ld={lda},{ldb}"
    msg = f"[{tag}] v13 == v11: {same13}  v14 == v11: {same14}  (v11 vs fp32 matmul max err {e:.3e})"
    if not same13:
        d = (outs[13].float() - outs[11].float()).abs()
        nz = torch.nonzero(d)
        msg += f"  **MISMATCH** {len(nz)} elements, max {float(d.max()):.3e}, first {nz[:4].tolist()}, rows bad {sorted(set((nz[:, 0] // 16).tolist()))[:24]} cols bad {sorted(set((nz[:, 1] // 16).tolist()))[:24]}"
    if not same14:
        d = (outs[14].float() - outs[11].float()).abs()
        nz = torch.nonzero(d)
        msg += f"  **MISMATCH(14)** {len(nz)} elements, max {float(d.max()):.3e}, first {nz[:4].tolist()}"
    print(msg, flush=True)
    return same13 and same14


def timeit(fn, it=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def bench(name, M, N, K, variants, rounds=2, out_f32=False, accumulate=False):
    a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    b = (torch.randn(N, K, device=DEV) * 0.5).bfloat16()
    c = torch.zeros(M, N, device=DEV, dtype=torch.float32 if out_f32 else torch.bfloat16)
    res = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            ms = timeit(lambda: ops.gemm(a, b, out=c, variant=v, out_f32=out_f32, accumulate=accumulate))
            res[v].append(2.0 * M * N * K / ms / 1e9)
    print(f"[bench {name} {M}x{N}x{K}] " + "  ".join(f"v{v}: " + "/".join(f"{x:.0f}" for x in r) + " TF" for v, r in res.items()), flush=True)


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    variants = (11, 13)
    for i, x in enumerate(sys.argv):
        if x == "--variants":
            variants = tuple(int(t) for t in sys.argv[i + 1].split(","))
    ok = True
    if "--bench-only" in sys.argv:
        v = tuple(x for x in variants if x != 11) if "--no11" in sys.argv else variants
        bench("o", 32768, 4096, 4096, v)
        bench("gate_up", 32768, 28672, 4096, v)
        bench("down", 32768, 4096, 14336, v)
        bench("dW_gate_up", 28672, 4096, 32768, v)
        sys.exit(0)
    ok &= case(256, 256, 256)
    ok &= case(256, 256, 512)
    ok &= case(512, 768, 1024)
    ok &= case(2048, 2304, 256)                                 # 72 tiles: every XCD's range, one tile per workgroup
    ok &= case(8192, 4096, 512)                                 # 512 tiles: two per workgroup (the persistent hand-over)
    ok &= case(8448, 2560, 384)                                 # 330 tiles: uneven ranges (some workgroups run a second tile, some do not)
    ok &= case(300, 520, 384)                                   # ragged M and N
    ok &= case(1000, 264, 256, lda=320, ldb=264)                # leading dimensions > K
    ok &= case(515, 1030, 640, bias=True)
    ok &= case(512, 512, 256, bias=True, gelu="erf")
    ok &= case(512, 512, 256, bias=True, gelu="tanh")
    ok &= case(768, 512, 256, residual=True)
    ok &= case(768, 512, 256, residual=True, res_row_mod=256)
    ok &= case(512, 512, 384, accumulate=True)
    ok &= case(512, 520, 384, accumulate=True, out_f32=True)
    ok &= case(257, 513, 256, out_f32=True)
    ok &= case(512, 1028, 256)                                  # N % 8 != 0: the scalar tail
    print("ALL CASES OK" if ok else "SOME CASES FAILED", flush=True)
    if not quick:
        bench("qkv", 32768, 6144, 4096, variants)
        bench("o", 32768, 4096, 4096, variants)
        bench("gate_up", 32768, 28672, 4096, variants)
        bench("down", 32768, 4096, 14336, variants)
        bench("dW_gate_up", 28672, 4096, 32768, variants, out_f32=False)
        bench("dW_down", 4096, 14336, 32768, variants)
        bench("lm_head", 8192, 128256, 4096, variants)
