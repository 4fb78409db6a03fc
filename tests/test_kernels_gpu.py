"""Kernel-level parity: every libmm355.so entry point (through the C ABI) against the CPU oracle
(oracle/ref_ops.py) on seeded inputs.  Needs an MI355X:  pytest -m gpu

Tolerances: the kernels compute in bf16 with fp32 accumulation; the oracle is evaluated in fp32 on the
SAME bf16-rounded inputs, so the only differences are the final bf16 rounding (rel 2^-8 = 3.9e-3) and
accumulation order.  Integer / index work (gathers, transposes, im2col) is compared bit-exactly.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402  (the checker)

DEV = "cuda"
# kernel generations no product path selects (GEMM variants 3-6, 8, 10, 12-14, attention variant 3) exist only in a library built with
# MM355_LEGACY_VARIANTS=1; their cases are generated only when MM355_TEST_LEGACY=1 asks for them (the default -m gpu run tests what ships)
LEGACY = os.environ.get("MM355_TEST_LEGACY") == "1"


@pytest.fixture(scope="module")
def ops():
    from metamorph_amd import ops as _ops
    from metamorph_amd import lib
    lib.load()
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def close(got, ref, rtol, atol, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        # a transposed result is the classic MFMA-layout bug: say so if that is what happened
        hint = ""
        if got.dim() == 2 and got.shape[0] == got.shape[1] and (got.t() - ref).abs().max() < atol + rtol * ref.abs().max():
            hint = " [matches the TRANSPOSE of the reference]"
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max err {err.max():.4g} "
                             f"(ref max {ref.abs().max():.4g}); first bad {idx} got {got[tuple(idx)]:.5g} ref {ref[tuple(idx)]:.5g}{hint}")


# ------------------------------------------------------------------------------------------------ GEMM

GEMM_SHAPES = [(128, 128, 64), (256, 256, 128), (300, 200, 192), (512, 384, 1152), (64, 130, 64), (1458, 1152, 4304),
               (200, 1152, 592), (1024, 1024, 4096)]


@pytest.mark.parametrize("variant", [0, 1, 2, 7, 9, 11] + ([3, 4, 5, 6, 8, 10] if LEGACY else []))
@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_gemm_plain(ops, variant, shape):
    M, N, K = shape
    if K % 64 and variant in (2, 4, 6, 7, 8, 9, 10, 11):
        pytest.skip("LDS-DMA variants need K % 64 == 0")
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = a.float() @ b.float().t()
    out = ops.gemm(a.to(DEV), b.to(DEV), variant=variant)
    close(out, ref, 1e-2, 0.02 * math.sqrt(K), f"gemm v{variant} {shape}")


@pytest.mark.parametrize("variant", [0, 1, 7, 11])
@pytest.mark.parametrize("shape", [(512, 384, 1152), (1024, 1024, 4096), (2304, 2048, 4096), (300, 200, 192), (256, 256, 8192)])
def test_gemm_fp32_output_tight(ops, variant, shape):
    """OUT_F32 removes the bf16 output rounding, so the kernel is compared with an fp64 product of the same bf16 operands to
    accumulation-order accuracy: a dropped / doubled k element (|a||b| ~ 1) or a mis-ordered fragment would be 10^4 x the bound."""
    M, N, K = shape
    if K % 64 and variant in (7, 11):
        pytest.skip("LDS-DMA variants need K % 64 == 0")
    a, b = rnd(M, K, seed=3), rnd(N, K, seed=4)
    ref = (a.double() @ b.double().t())
    out = ops.gemm(a.to(DEV), b.to(DEV), out_f32=True, variant=variant).double().cpu()
    # fp32 accumulation of exact bf16 products, |a| ~ |b| ~ 1: partial sums ~ sqrt(k), error random-walks to ~ 2^-24 K / 2.4 (1e-4 at
    # K = 4096); 4 K 2^-24 (1e-3 at K = 4096) is ~10 sigma and still 500x below one missing product
    bound = 4 * K * 2.0 ** -24
    bad = (out - ref).abs() > bound
    assert not bool(bad.any()), f"gemm f32 v{variant} {shape}: {int(bad.sum())} elements beyond the fp32 accumulation bound; max err {float((out - ref).abs().max()):.3e}"
    # the bf16 output of the same launch is that fp32 value rounded once (<= 1/2 ulp = 2^-9 relative)
    o16 = ops.gemm(a.to(DEV), b.to(DEV), variant=variant).double().cpu()
    assert bool(((o16 - ref).abs() <= 2.0 ** -8 * ref.abs() + bound).all()), f"gemm bf16 v{variant} {shape}: more than one rounding away"


@pytest.mark.parametrize("shape", [(256, 256, 128), (700, 520, 256), (1024, 768, 2048), (2304, 2048, 1152), (4096, 4096, 8192)])
def test_gemm_pingpong_race_screen(ops, shape):
    """The staggered two-group 256x256 kernel (variant 11): counted waits + barriers are the only ordering between the LDS-DMA
    writes and the fragment reads, so screen it repeatedly (different data each round) against the plain 128x128 kernel, which
    is bit-comparable (same bf16 products, fp32 accumulation in the same K order per 32-wide MFMA step)."""
    M, N, K = shape
    for rep in range(4):
        a, b = rnd(M, K, seed=10 + rep).to(DEV), rnd(N, K, seed=20 + rep).to(DEV)
        out = ops.gemm(a, b, variant=11)
        ref = ops.gemm(a, b, variant=1)
        assert torch.equal(out, ref), f"variant 11 differs from variant 1 at {shape} round {rep}: max |d| = {float((out.float() - ref.float()).abs().max())}"
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    close(ops.gemm(a.to(DEV), b.to(DEV), variant=11), a.float() @ b.float().t(), 1e-2, 0.02 * math.sqrt(K), f"gemm v11 {shape}")


if LEGACY:
    @pytest.mark.parametrize("shape", [(256, 256, 256), (300, 520, 384), (2048, 2304, 256), (8192, 4096, 512), (8448, 2560, 384), (1024, 1028, 1152)])
    def test_gemm_one_wave_per_simd_stream_is_bit_identical(ops, shape):
        """Variant 13 (csrc/gemm_st.hip: persistent workgroups of four waves, 128 x 128 wave tiles with all 256 accumulators in AGPRs, a
        hand-placed instruction stream) and variant 14 (the same stream serialised: every LDS read / LDS-DMA piece waited for on the spot)
        against the eight-wave ping-pong kernel: the same 16x16x32 MFMA per 32 k in ascending order and the same fused store, so every
        output -- ragged edges, the N % 8 scalar tail, several tiles per workgroup (the persistent hand-over with the next tile's first two
        K stages in flight during the epilogue), uneven per-XCD tile ranges, each epilogue flag -- is BIT-identical; two data sets (race screen)."""
        M, N, K = shape
        for rep in range(2):
            a, b = rnd(M, K, seed=30 + rep).to(DEV), rnd(N, K, seed=40 + rep).to(DEV)
            ref = ops.gemm(a, b, variant=11)
            for v in (13, 14):
                out = ops.gemm(a, b, variant=v)
                assert torch.equal(out, ref), f"variant {v} differs from variant 11 at {shape} round {rep}: max |d| = {float((out.float() - ref.float()).abs().max())}"
        a, b = rnd(M, K, seed=1, scale=0.3).to(DEV), rnd(N, K, seed=2, scale=0.3).to(DEV)
        bias, res, c0 = rnd(N, seed=5).to(DEV), rnd(64, N, seed=6).to(DEV), rnd(M, N, seed=8).to(DEV)
        for kw in (dict(bias=bias, gelu="erf"), dict(bias=bias, gelu="tanh"), dict(bias=bias, residual=res, res_row_mod=64), dict(accumulate=True),
                   dict(accumulate=True, out_f32=True)):
            outs = []
            for v in (11, 13, 14):
                c = (c0.float() if kw.get("out_f32") else c0).clone()
                ops.gemm(a, b, out=c, variant=v, **kw)
                outs.append(c)
            assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0]), f"epilogue {sorted(kw)} at {shape}"


    def test_gemm_stream_variant_rejects_short_k(ops):
        from metamorph_amd.lib import Mm355Error
        a, b = rnd(256, 128, seed=1).to(DEV), rnd(256, 128, seed=2).to(DEV)
        with pytest.raises(Mm355Error):
            ops.gemm(a, b, variant=13)                              # two K stages: the stream fetches two stages ahead (K >= 256, K % 128 == 0)


@pytest.mark.parametrize("shapes", [((512, 768, 256), (300, 520, 384)), ((256, 256, 128), (256, 256, 128)),
                                    ((1024, 2048, 1152), (2300, 264, 128)), ((8, 8, 128), (2048, 1280, 640))])
def test_gemm_pair_equals_two_launches(ops, shapes):
    """mm355_gemm_pair_bf16: two problems in one grid (workgroups [0, n0) / [n0, n0 + n1)) must give exactly what the same
    kernel gives launched per problem -- ragged edges, different K, accumulate and fp32 outputs, and a repeat (race screen)."""
    (M0, N0, K0), (M1, N1, K1) = shapes
    for rep in range(3):
        a0, b0 = rnd(M0, K0, seed=30 + rep).to(DEV), rnd(N0, K0, seed=40 + rep).to(DEV)
        a1, b1 = rnd(M1, K1, seed=50 + rep).to(DEV), rnd(N1, K1, seed=60 + rep).to(DEV)
        assert ops.gemm_pair_supported(a0, b0, a1, b1)
        o0 = torch.empty(M0, N0, device=DEV, dtype=torch.bfloat16)
        o1 = torch.empty(M1, N1, device=DEV, dtype=torch.bfloat16)
        ops.gemm_pair(a0, b0, o0, False, a1, b1, o1, False)
        r0, r1 = ops.gemm(a0, b0, variant=11), ops.gemm(a1, b1, variant=11)
        assert torch.equal(o0, r0) and torch.equal(o1, r1), f"pair differs from single launches at {shapes} round {rep}"
    close(o0, a0.float() @ b0.float().t(), 1e-2, 0.02 * math.sqrt(K0), "pair problem 0 vs fp32")
    close(o1, a1.float() @ b1.float().t(), 1e-2, 0.02 * math.sqrt(K1), "pair problem 1 vs fp32")
    # accumulate into bf16 (problem 0) and fp32 (problem 1) targets
    c0 = rnd(M0, N0, seed=70).to(DEV)
    c1 = rnd(M1, N1, seed=71).to(DEV).float()
    e0, e1 = c0.clone(), c1.clone()
    ops.gemm_pair(a0, b0, c0, True, a1, b1, c1, True)
    ops.gemm(a0, b0, out=e0, accumulate=True, variant=11)
    ops.gemm(a1, b1, out=e1, accumulate=True, variant=11)
    assert torch.equal(c0, e0) and torch.equal(c1, e1)


def test_gemm_pingpong_operand_beyond_2gib(ops):
    """Row-major operands are addressed from the tile's first row, so a 2 GiB+ operand stays on the ping-pong kernel: row
    windows at the start, in the middle and past the 2 GiB mark must equal the plain kernel run on just those rows -- as the
    A operand (tall activations) and as the B operand."""
    M, N, K = (1 << 20) + 300, 256, 1024                     # A: 2.0006 GiB
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    b = rnd(N, K, seed=5).to(DEV)
    assert a.numel() * 2 > 2 ** 31
    out = ops.gemm(a, b, variant=11)
    for lo in (0, M // 2 - 128, M - 812):
        ref = ops.gemm(a[lo:lo + 812], b, variant=1)
        assert torch.equal(out[lo:lo + 812], ref), f"rows {lo}..{lo + 812}"
    del out
    out_t = ops.gemm(b, a, variant=11)                       # [256, M]: the big matrix as B
    for lo in (0, M - 812):
        ref = ops.gemm(b, a[lo:lo + 812], variant=1)
        assert torch.equal(out_t[:, lo:lo + 812], ref), f"cols {lo}..{lo + 812}"


def test_gemm_pair_rejects_ineligible(ops):
    from metamorph_amd.lib import Mm355Error
    a, b = rnd(256, 192, seed=1).to(DEV), rnd(256, 192, seed=2).to(DEV)          # K = 192: not whole pairs of K tiles
    a1, b1 = rnd(256, 128, seed=3).to(DEV), rnd(256, 128, seed=4).to(DEV)
    assert not ops.gemm_pair_supported(a, b, a1, b1)
    o, o1 = torch.empty(256, 256, device=DEV, dtype=torch.bfloat16), torch.empty(256, 256, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(Mm355Error):
        ops.gemm_pair(a, b, o, False, a1, b1, o1, False)


@pytest.mark.parametrize("variant", [1, 2, 7, 11] + ([6, 10] if LEGACY else []))
def test_gemm_epilogues(ops, variant):
    M, N, K = 320, 256, 128
    a, b = rnd(M, K, seed=3, scale=0.3), rnd(N, K, seed=4, scale=0.3)
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    acc = a.float() @ b.float().t()
    tol = dict(rtol=1e-2, atol=0.03)
    close(ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), variant=variant), acc + bias.float(), what="bias", **tol)
    close(ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), gelu="erf", variant=variant), R.gelu_erf(acc + bias.float()), what="gelu_erf", **tol)
    close(ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), gelu="tanh", variant=variant), R.gelu_tanh(acc + bias.float()), what="gelu_tanh", **tol)
    close(ops.gemm(a.to(DEV), b.to(DEV), residual=res.to(DEV), variant=variant), acc + res.float(), what="residual", **tol)
    pos = rnd(64, N, seed=7)                                   # residual indexed by row % 64 (position embedding)
    close(ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=pos.to(DEV), res_row_mod=64, variant=variant),
          acc + bias.float() + pos.float().repeat(5, 1), what="bias+pos", **tol)
    c0 = rnd(M, N, seed=8)
    c = c0.to(DEV).clone()
    ops.gemm(a.to(DEV), b.to(DEV), out=c, accumulate=True, variant=variant)
    close(c, acc + c0.float(), what="accumulate", **tol)
    cf = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    ops.gemm(a.to(DEV), b.to(DEV), out=cf, variant=variant)
    close(cf, acc, rtol=1e-4, atol=1e-3, what="out_f32")
    ops.gemm(a.to(DEV), b.to(DEV), out=cf, accumulate=True, variant=variant)
    close(cf, 2 * acc, rtol=1e-4, atol=2e-3, what="out_f32 accumulate")


def test_gemm_strided_views(ops):
    M, N, K = 192, 136, 128
    big_a, big_b = rnd(M, 3 * K, seed=9), rnd(N, 2 * K, seed=10)
    a, b = big_a[:, K:2 * K], big_b[:, K:]
    ref = a.float() @ b.float().t()
    out_big = torch.zeros(M, 256, device=DEV, dtype=torch.bfloat16)
    ops.gemm(big_a.to(DEV)[:, K:2 * K], big_b.to(DEV)[:, K:], out=out_big[:, 64:64 + N])
    close(out_big[:, 64:64 + N], ref, 1e-2, 0.25, "strided gemm")
    assert float(out_big[:, :64].abs().max()) == 0 and float(out_big[:, 64 + N:].abs().max()) == 0


@pytest.mark.parametrize("shape", [(256, 256, 64), (512, 768, 256), (264, 136, 128), (4096, 1152, 1024), (1152, 4096, 448), (8, 16, 64)])
def test_gemm_tn(ops, shape):
    """dW form on untransposed operands: C[M,N] = At[K,M]^T Bt[K,N] (ds_read_b64_tr_b16 fragment gather)."""
    M, N, K = shape
    at, bt = rnd(K, M, seed=11, scale=0.5), rnd(K, N, seed=12, scale=0.5)
    ref = at.float().t() @ bt.float()
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm_tn(at.to(DEV), bt.to(DEV), out)
    close(out, ref, 1e-2, 0.02 * math.sqrt(K), f"gemm_tn {shape}")
    c0 = rnd(M, N, seed=13)
    c = c0.to(DEV).clone()
    ops.gemm_tn(at.to(DEV), bt.to(DEV), c, accumulate=True)
    close(c, ref + c0.float(), 1e-2, 0.02 * math.sqrt(K) + 0.03, "gemm_tn accumulate")
    cf = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    ops.gemm_tn(at.to(DEV), bt.to(DEV), cf)
    close(cf, ref, 1e-4, 2e-3, "gemm_tn f32")
    # strided operands (column blocks of wider activations)
    big = rnd(K, M + 64, seed=14, scale=0.5)
    ops.gemm_tn(big.to(DEV)[:, 32:32 + M] if M % 8 == 0 else big.to(DEV)[:, :M], bt.to(DEV), out)
    close(out, big[:, 32:32 + M].float().t() @ bt.float(), 1e-2, 0.02 * math.sqrt(K), "gemm_tn strided A")


@pytest.mark.parametrize("shape", [(256, 256, 128), (520, 264, 256), (1024, 776, 1024), (2304, 2048, 2048)])
def test_gemm_contraction_major_pingpong(ops, shape):
    """TN (weight gradient) and NN (input gradient) forms of the ping-pong kernel against the NT kernel on explicitly
    transposed copies: same bf16 products and fp32 accumulation order => bit-identical; repeated to screen for races."""
    M, N, K = shape
    for rep in range(3):
        at, bt = rnd(K, M, seed=30 + rep, scale=0.5).to(DEV), rnd(K, N, seed=40 + rep, scale=0.5).to(DEV)
        ref = ops.gemm(at.t().contiguous(), bt.t().contiguous(), variant=1)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_tn(at, bt, out)
        assert torch.equal(out, ref), f"gemm_tn (ping-pong) {shape} round {rep}: max |d| = {float((out.float() - ref.float()).abs().max())}"
        a = rnd(M, K, seed=50 + rep, scale=0.5).to(DEV)
        res = rnd(M, N, seed=60 + rep).to(DEV)
        ref = ops.gemm(a, bt.t().contiguous(), residual=res, variant=1)
        out = ops.gemm_nn(a, bt, residual=res)
        assert torch.equal(out, ref), f"gemm_nn (ping-pong) {shape} round {rep}: max |d| = {float((out.float() - ref.float()).abs().max())}"
    close(ops.gemm_nn(a, bt), a.float().cpu() @ bt.float().cpu(), 1e-2, 0.02 * math.sqrt(K), f"gemm_nn {shape}")
    wide = rnd(K, N + 64, seed=70, scale=0.5).to(DEV)             # strided weight view (column block of a fused buffer)
    close(ops.gemm_nn(a, wide[:, 32:32 + N]), a.float().cpu() @ wide[:, 32:32 + N].float().cpu(), 1e-2, 0.02 * math.sqrt(K), "gemm_nn strided")


def test_gemm_nn_rejects_ragged_k(ops):
    from metamorph_amd.lib import Mm355Error
    with pytest.raises(Mm355Error):
        ops.gemm_nn(rnd(64, 192, seed=1).to(DEV), rnd(192, 64, seed=2).to(DEV))


def test_gemm_tn_rejects_ragged_k(ops):
    from metamorph_amd.lib import Mm355Error
    at, bt = rnd(100, 64, seed=1).to(DEV), rnd(100, 64, seed=2).to(DEV)
    with pytest.raises(Mm355Error):
        ops.gemm_tn(at, bt, torch.empty(64, 64, device=DEV, dtype=torch.bfloat16))


@pytest.mark.parametrize("M,h,Rp", [(300, 512, 304), (1024, 4096, 1024), (777, 1152, 896)])
def test_rmsnorm_apply_t_equals_transposed_forward(ops, M, h, Rp):
    """y^T from the saved rstd in one pass == transpose(rmsnorm_fwd(x)) bit for bit; padding columns are zero; rstd vs oracle."""
    x = rnd(M, h, seed=21, scale=1.5).to(DEV)
    w = (1.0 + 0.1 * rnd(h, seed=22)).bfloat16().to(DEV)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5, want_rstd=True)
    assert torch.equal(y, ops.rmsnorm_fwd(x, w, 1e-5))
    ref_rstd = torch.rsqrt((x.float() ** 2).mean(-1) + 1e-5)
    close(rstd.cpu(), ref_rstd.cpu(), 1e-5, 1e-6, "rstd")
    yt = ops.rmsnorm_apply_t(x, w, rstd, Rp)
    assert yt.shape == (h, Rp)
    assert torch.equal(yt[:, :M], y.t())
    assert not yt[:, M:].any()


def test_grad_norm_and_bias_grad_sums_are_deterministic(ops):
    """mm355_sumsq_bf16 / mm355_colsum_bf16 use no atomics: repeated launches give the same bits, and both ACCUMULATE into `out`."""
    x = (torch.randn(37_000_003, device=DEV) * 0.3).bfloat16()          # > MM355_SUMSQ_PARTIALS workgroups, ragged tail
    outs = []
    for _ in range(4):
        s = torch.zeros(1, device=DEV)
        ops.sumsq_(x, s)
        outs.append(s.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = float((x.double() ** 2).sum())
    assert abs(float(outs[0]) - ref) <= 2e-5 * ref
    s2 = outs[0].clone()
    ops.sumsq_(x[:1000], s2)                                           # accumulates
    assert abs(float(s2) - float(outs[0]) - float((x[:1000].double() ** 2).sum())) <= 1e-5 * ref
    y = (torch.randn(4099, 1160, device=DEV) * 0.5).bfloat16()
    cs = []
    for _ in range(3):
        c = torch.zeros(1160, device=DEV)
        ops.colsum_f32(y, c)
        cs.append(c.clone())
    assert torch.equal(cs[0], cs[1]) and torch.equal(cs[0], cs[2])
    close(cs[0].cpu(), y.float().sum(0).cpu(), 1e-4, 1e-2, "colsum big")
    c2 = cs[0].clone()
    ops.colsum_f32(y, c2)
    close(c2.cpu(), 2 * y.float().sum(0).cpu(), 1e-4, 2e-2, "colsum accumulates")


def test_transpose_and_colsum(ops):
    for (r, c) in [(64, 64), (200, 136), (729, 1152), (130, 72)]:
        x = rnd(r, c, seed=r)
        t = ops.transpose(x.to(DEV))
        assert torch.equal(t.cpu(), x.t()), (r, c)
    x = rnd(515, 264, seed=11)
    s = torch.zeros(264, device=DEV)
    ops.colsum_f32(x.to(DEV), s)
    close(s, x.float().sum(0), 1e-4, 1e-3, "colsum")
    # the narrow-load form (few rows, or a column count that is not a multiple of 8) and a strided column block of a wider matrix
    for (r, c) in [(100, 264), (600, 203)]:
        x = rnd(r, c, seed=r + c)
        s = torch.zeros(c, device=DEV)
        ops.colsum_f32(x.to(DEV), s)
        close(s, x.float().sum(0), 1e-4, 1e-3, f"colsum {r}x{c}")
    wide = rnd(777, 3 * 1152, seed=5).to(DEV)
    blk = wide[:, 1152:2 * 1152]
    s = torch.zeros(1152, device=DEV)
    ops.colsum_f32(blk, s)
    close(s, blk.float().sum(0), 1e-4, 2e-3, "colsum of a column block")


# ------------------------------------------------------------------------------------------------ norms

@pytest.mark.parametrize("h", [256, 1152, 4096])
def test_rmsnorm_fwd_bwd(ops, h):
    M = 37
    x, w, dy, dres = rnd(M, h, seed=1), (1 + 0.1 * rnd(h, seed=2).float()).bfloat16(), rnd(M, h, seed=3), rnd(M, h, seed=4)
    y = ops.rmsnorm_fwd(x.to(DEV), w.to(DEV), 1e-5)
    yr = R.rmsnorm(x, w, 1e-5)
    close(y, yr, 8e-3, 1e-2, "rmsnorm fwd")
    xf = x.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    R.rmsnorm(xf, wf, 1e-5).backward(dy.float())
    for atomic in (False, True):                                # two-stage (workspace) and fp32-atomic weight gradient
        dw = torch.full((h,), 0.5, device=DEV)                  # accumulates onto what is there
        dx = ops.rmsnorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), 1e-5, dres=dres.to(DEV), dw_f32=dw, atomic=atomic)
        close(dx, xf.grad + dres.float(), 1e-2, 2e-2, "rmsnorm dx")
        close(dw - 0.5, wf.grad, 1e-2, 5e-2, f"rmsnorm dw atomic={atomic}")


def test_rmsnorm_bwd_large_deterministic(ops):
    M, h = 9000, 4096                                           # several rows per workgroup, ragged last workgroup
    x, w, dy = rnd(M, h, seed=1).to(DEV), (1 + 0.1 * rnd(h, seed=2).float()).bfloat16().to(DEV), rnd(M, h, seed=3).to(DEV)
    outs = []
    for _ in range(2):
        dw = torch.zeros(h, device=DEV)
        ops.rmsnorm_bwd(dy, x, w, 1e-5, dw_f32=dw)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    xh = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5)).bfloat16().float()
    close(outs[0], (dy.float() * xh).sum(0), 1e-3, 1e-3 * math.sqrt(M), "rmsnorm dw two-stage")


def test_layernorm_fwd(ops):
    x, w, b = rnd(50, 1152, seed=1), (1 + 0.1 * rnd(1152, seed=2).float()).bfloat16(), rnd(1152, seed=3, scale=0.1)
    y = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6)
    close(y, R.layernorm(x.float(), w.float(), b.float(), 1e-6), 8e-3, 1e-2, "layernorm")


@pytest.mark.parametrize("Mh", [(50, 1152), (9000, 1152), (37, 256)])
def test_layernorm_bwd(ops, Mh):
    M, h = Mh
    x, dy, dres = rnd(M, h, seed=1), rnd(M, h, seed=2), rnd(M, h, seed=3)
    w, b = (1 + 0.1 * rnd(h, seed=4).float()).bfloat16(), rnd(h, seed=5, scale=0.1)
    xf, wf, bf = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    R.layernorm(xf, wf, bf, 1e-6).backward(dy.float())
    dw, db = torch.full((h,), 0.25, device=DEV), torch.full((h,), -0.5, device=DEV)      # accumulate onto what is there
    dx = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), 1e-6, dw, db, dres=dres.to(DEV))
    close(dx, xf.grad + dres.float(), 1e-2, 2e-2, "layernorm dx")
    close(dw - 0.25, wf.grad, 1e-3, 1e-3 * math.sqrt(M) + 1e-3, "layernorm dw")
    close(db + 0.5, bf.grad, 1e-3, 1e-3 * math.sqrt(M) + 1e-3, "layernorm db")
    dw2, db2 = torch.full((h,), 0.25, device=DEV), torch.full((h,), -0.5, device=DEV)
    ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), 1e-6, dw2, db2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                                 # fixed summation order


# ------------------------------------------------------------------------------------------------ rope

def test_rope(ops):
    B, L, Hq, Hkv, d = 2, 70, 4, 2, 128
    cos, sin = ops.rope_table(L, d, 500000.0, DEV)
    pos = torch.arange(L)[None]
    cr, sr = R.rope_tables(pos, d, 500000.0, torch.bfloat16)
    close(cos, cr[0], 0, 8e-3, "rope cos")
    close(sin, sr[0], 0, 8e-3, "rope sin")
    ld = (Hq + 2 * Hkv) * d
    qkv = rnd(B * L, ld, seed=5)
    dev = qkv.to(DEV).clone()
    ops.rope_qk_(dev, B, L, Hq, Hkv, d, cos, sin)
    q = qkv[:, :Hq * d].view(B, L, Hq, d).transpose(1, 2)
    k = qkv[:, Hq * d:(Hq + Hkv) * d].view(B, L, Hkv, d).transpose(1, 2)
    cb, sb = cos.cpu()[None].expand(B, L, d), sin.cpu()[None].expand(B, L, d)
    qe = R.rope_apply(q, cb, sb).transpose(1, 2).reshape(B * L, Hq * d)
    ke = R.rope_apply(k, cb, sb).transpose(1, 2).reshape(B * L, Hkv * d)
    close(dev[:, :Hq * d], qe, 8e-3, 8e-3, "rope q")
    close(dev[:, Hq * d:(Hq + Hkv) * d], ke, 8e-3, 8e-3, "rope k")
    assert torch.equal(dev[:, (Hq + Hkv) * d:].cpu(), qkv[:, (Hq + Hkv) * d:]), "v block must be untouched"
    # inverse is the transpose of the rotation: <R x, y> == <x, R^T y>
    y = rnd(B * L, ld, seed=6)
    ydev = y.to(DEV).clone()
    ops.rope_qk_(ydev, B, L, Hq, Hkv, d, cos, sin, inverse=True)
    n = (Hq + Hkv) * d
    lhs = (dev[:, :n].float().cpu() * y[:, :n].float()).sum()
    rhs = (qkv[:, :n].float() * ydev[:, :n].float().cpu()).sum()
    assert abs(float(lhs - rhs)) < 2e-2 * max(1.0, abs(float(lhs))), (float(lhs), float(rhs))


def test_rope_table_freq_matches_hf_recorded_tables(ops):
    """mm355_rope_table_freq (scaled RoPE: LLaMA-3.1 "llama3", 3.2, linear, and the default through the same entry point) against the cos / sin
    rows HF's own LlamaRotaryEmbedding produced (tests/golden/r6_rope_tables.npz), positions up to 4095: at most one bf16 step anywhere
    (device vs host cosf), and bit-identical on >= 99.5 % of the entries."""
    import json
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "r6_rope_tables.npz"))
    pos = torch.from_numpy(g["positions"])
    for name in sorted({k.split("::")[0] for k in g.files if "::" in k}):
        c = json.loads(str(g[f"{name}::cfg"]))
        d = c["head_dim"]
        cos, sin = ops.rope_table_freq(4096, d, g[f"{name}::inv_freq"], float(g[f"{name}::attention_scaling"]), DEV)
        assert cos.shape == (4096, d) and torch.equal(cos[:, :d // 2], cos[:, d // 2:]) and torch.equal(sin[:, :d // 2], sin[:, d // 2:])
        for got, key in ((cos, "cos"), (sin, "sin")):
            want = torch.from_numpy(g[f"{name}::{key}_bf16"])
            have = got.cpu()[pos].float()
            close(have, want, 0, 8e-3, f"{name} {key}")
            same = float((have == want).float().mean())
            assert same >= 0.995, (name, key, same)
    # the theta entry point (mm355_rope_table) builds the default table on the device from theta alone: same table
    cd, sd_ = ops.rope_table(4096, 128, 500000.0, DEV)
    cf, sf = ops.rope_table_freq(4096, 128, g["default::inv_freq"], 1.0, DEV)
    assert float((cd != cf).float().mean()) < 5e-3 and float((cd.float() - cf.float()).abs().max()) <= 8e-3
    assert float((sd_ != sf).float().mean()) < 5e-3


# ------------------------------------------------------------------------------------------------ attention

ATT_CASES = [  # B, L, Hq, Hkv, d, causal, seqlens
    (2, 200, 4, 2, 128, True, [200, 137]),
    (1, 64, 2, 1, 128, True, None),
    (2, 130, 4, 4, 64, True, [130, 5]),
    (2, 100, 3, 3, 72, False, None),
    (1, 729, 2, 2, 72, False, None),
    (2, 333, 8, 2, 128, True, [333, 256]),
    (1, 513, 4, 2, 128, True, None),                    # d == 128 LDS-DMA kernels: several 128-row query blocks, ragged tail
    (2, 200, 2, 2, 128, False, [200, 77]),
    (2, 256, 4, 1, 128, True, [1, 256]),
    (1, 300, 6, 3, 128, True, None),                    # block order: groups of 2 heads walked fastest; dK/dV: odd KV-head count
    (1, 260, 5, 5, 128, True, None),                    # no GQA, odd head count: one head at a time
    (2, 384, 12, 4, 128, True, [300, 384]),             # groups of 3
    (2, 320, 16, 2, 128, True, [320, 191]),             # GQA 8:1 at d = 128 (LLaMA-3-70B: 64 query / 8 KV heads), ragged
]


def _attn_setup(case, seed=0):
    B, L, Hq, Hkv, d, causal, seqlens = case
    ld = (Hq + 2 * Hkv) * d
    qkv = rnd(B * L, ld, seed=seed, scale=0.7)
    q = qkv[:, :Hq * d].view(B, L, Hq, d).transpose(1, 2)
    k = qkv[:, Hq * d:(Hq + Hkv) * d].view(B, L, Hkv, d).transpose(1, 2)
    v = qkv[:, (Hq + Hkv) * d:].view(B, L, Hkv, d).transpose(1, 2)
    valid = None
    if seqlens is not None:
        valid = torch.arange(L)[None] < torch.tensor(seqlens)[:, None]
    return qkv, q, k, v, valid


@pytest.mark.parametrize("case", ATT_CASES)
def test_attn_fwd(ops, case):
    B, L, Hq, Hkv, d, causal, seqlens = case
    qkv, q, k, v, valid = _attn_setup(case)
    ref = R.attention(q, k, v, valid, causal=causal).transpose(1, 2).reshape(B, L, Hq * d)
    dev = qkv.to(DEV)
    sl = torch.tensor(seqlens, dtype=torch.int32, device=DEV) if seqlens else None
    o, lse = ops.attn_fwd(dev[:, :Hq * d], dev[:, Hq * d:(Hq + Hkv) * d], dev[:, (Hq + Hkv) * d:], B, L, Hq, Hkv, d, d ** -0.5, causal, sl)
    o = o.view(B, L, Hq * d)
    for b in range(B):
        n = seqlens[b] if seqlens else L
        close(o[b, :n], ref[b, :n], 1e-2, 1e-2, f"attn fwd {case} sample {b}")
        assert float(o[b, n:].abs().max()) == 0 if n < L else True
    # lse against the oracle definition
    s = torch.matmul(q.float(), k.float().repeat_interleave(Hq // Hkv, 1).transpose(-1, -2)) * d ** -0.5
    if causal:
        s = s.masked_fill(~torch.ones(L, L, dtype=torch.bool).tril(), float("-inf"))
    if valid is not None:
        s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    lse_ref = torch.logsumexp(s, -1)
    for b in range(B):
        n = seqlens[b] if seqlens else L
        close(lse[b, :, :n], lse_ref[b, :, :n], 1e-3, 1e-2, "lse")


@pytest.mark.parametrize("case", ATT_CASES)
def test_attn_bwd(ops, case):
    B, L, Hq, Hkv, d, causal, seqlens = case
    qkv, q, k, v, valid = _attn_setup(case, seed=3)
    do = rnd(B * L, Hq * d, seed=4, scale=0.5)
    if valid is not None:                                       # padded query rows carry no gradient
        do = (do.view(B, L, -1) * valid[:, :, None]).reshape(B * L, -1).contiguous()
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = R.attention(qf, kf, vf, valid, causal=causal)
    # nan-safe: padded rows under causal masking are finite in the oracle as well
    (ref * do.float().view(B, L, Hq, d).transpose(1, 2)).sum().backward()
    dev = qkv.to(DEV)
    sl = torch.tensor(seqlens, dtype=torch.int32, device=DEV) if seqlens else None
    qd, kd, vd = dev[:, :Hq * d], dev[:, Hq * d:(Hq + Hkv) * d], dev[:, (Hq + Hkv) * d:]
    o, lse = ops.attn_fwd(qd, kd, vd, B, L, Hq, Hkv, d, d ** -0.5, causal, sl)
    dqkv = torch.full_like(dev, float("nan"))                    # every element must be written
    ops.attn_bwd(qd, kd, vd, o, do.to(DEV), lse, B, L, Hq, Hkv, d, d ** -0.5, causal, sl,
                 dqkv[:, :Hq * d], dqkv[:, Hq * d:(Hq + Hkv) * d], dqkv[:, (Hq + Hkv) * d:])
    tol = dict(rtol=2e-2, atol=2e-2)
    close(dqkv[:, :Hq * d].view(B, L, Hq, d), qf.grad.transpose(1, 2), what=f"dq {case}", **tol)
    close(dqkv[:, Hq * d:(Hq + Hkv) * d].view(B, L, Hkv, d), kf.grad.transpose(1, 2), what=f"dk {case}", **tol)
    close(dqkv[:, (Hq + Hkv) * d:].view(B, L, Hkv, d), vf.grad.transpose(1, 2), what=f"dv {case}", **tol)


@pytest.mark.parametrize("L,Hq", [(2048, 32), (4096, 32), (4096, 64)])
def test_attn3_full_length_against_oracle_slice(ops, L, Hq):
    """The d = 128 LDS-DMA kernels at BASELINE sequence lengths (configs[1] L = 2048, configs[2] L = 4096), LLaMA-3-8B head
    geometry (32 query / 8 KV heads) and configs[4]'s LLaMA-3-70B geometry (64 query / 8 KV heads: eight query heads per KV group,
    L = 4096), per-sample lengths: forward output, lse and all three gradients of ONE (sample, KV group) slice against the fp32
    oracle (4 or 8 query heads x L x L scores on the CPU), for a full-length and a ragged sample."""
    B, Hkv, d = 2, 8, 128
    seqlens = [L, L - 333]
    ld = (Hq + 2 * Hkv) * d
    qkv = rnd(B * L, ld, seed=11 + L, scale=0.7)
    do = rnd(B * L, Hq * d, seed=12 + L, scale=0.5)
    valid = torch.arange(L)[None] < torch.tensor(seqlens)[:, None]
    do = (do.view(B, L, -1) * valid[:, :, None]).reshape(B * L, -1).contiguous()
    dev = qkv.to(DEV)
    sl = torch.tensor(seqlens, dtype=torch.int32, device=DEV)
    qd, kd, vd = dev[:, :Hq * d], dev[:, Hq * d:(Hq + Hkv) * d], dev[:, (Hq + Hkv) * d:]
    o, lse = ops.attn_fwd(qd, kd, vd, B, L, Hq, Hkv, d, d ** -0.5, True, sl)
    dqkv = torch.full_like(dev, float("nan"))
    ops.attn_bwd(qd, kd, vd, o, do.to(DEV), lse, B, L, Hq, Hkv, d, d ** -0.5, True, sl,
                 dqkv[:, :Hq * d], dqkv[:, Hq * d:(Hq + Hkv) * d], dqkv[:, (Hq + Hkv) * d:])
    assert bool(torch.isfinite(dqkv.float()).all()), "every gradient element must be written"
    rep = Hq // Hkv
    for b, g in ((0, 5), (1, 2)):                                # (sample, KV head): query heads g*rep .. g*rep + rep - 1
        n = seqlens[b]
        rows = slice(b * L, b * L + L)
        q = qkv[rows, :Hq * d].view(L, Hq, d)[:, g * rep:(g + 1) * rep].permute(1, 0, 2)[None].float().requires_grad_(True)
        k = qkv[rows, Hq * d:(Hq + Hkv) * d].view(L, Hkv, d)[:, g:g + 1].permute(1, 0, 2)[None].float().requires_grad_(True)
        v = qkv[rows, (Hq + Hkv) * d:].view(L, Hkv, d)[:, g:g + 1].permute(1, 0, 2)[None].float().requires_grad_(True)
        ref = R.attention(q, k, v, valid[b:b + 1], causal=True)                          # [1, rep, L, d]
        dos = do[rows].view(L, Hq, d)[:, g * rep:(g + 1) * rep].permute(1, 0, 2)[None].float()
        (ref * dos).sum().backward()
        got_o = o[rows].view(L, Hq, d)[:, g * rep:(g + 1) * rep].permute(1, 0, 2)
        close(got_o[:, :n], ref[0, :, :n], 1e-2, 4e-3, f"attn3 fwd L={L} sample {b} group {g}")
        assert float(got_o[:, n:].float().abs().max()) == 0 if n < L else True
        s = torch.matmul(q[0].detach(), k[0].detach().transpose(-1, -2)) * d ** -0.5
        s = s.masked_fill(~torch.ones(L, L, dtype=torch.bool).tril(), float("-inf")).masked_fill(~valid[b][None, None, :], float("-inf"))
        close(lse[b, g * rep:(g + 1) * rep, :n], torch.logsumexp(s, -1)[:, :n], 1e-3, 5e-3, f"attn3 lse L={L}")
        tol = dict(rtol=2e-2, atol=1e-2)
        gq = dqkv[rows, :Hq * d].view(L, Hq, d)[:, g * rep:(g + 1) * rep].permute(1, 0, 2)
        gk = dqkv[rows, Hq * d:(Hq + Hkv) * d].view(L, Hkv, d)[:, g]
        gv = dqkv[rows, (Hq + Hkv) * d:].view(L, Hkv, d)[:, g]
        close(gq, q.grad[0], what=f"attn3 dq L={L} sample {b} group {g}", **tol)
        close(gk, k.grad[0, 0], what=f"attn3 dk L={L} sample {b} group {g}", **tol)
        close(gv, v.grad[0, 0], what=f"attn3 dv L={L} sample {b} group {g}", **tol)


def test_gemm_swiglu_fused_equals_two_launches(ops):
    """mm355_gemm_swiglu_bf16 (gate|up GEMM with SiLU(gate) * up formed in the epilogue; the B tile staged from two row ranges of the fused
    weight) writes bit for bit what mm355_gemm_bf16 + mm355_swiglu_fwd write: LLaMA-3-8B widths at 4096 rows, a ragged row count, and
    TinyLlama's I = 5632 (44 tiles of 128 channels); unsupported shapes are refused, not approximated."""
    for (M, K, I) in [(4096, 4096, 14336), (2304 + 77, 2048, 5632), (6000, 1024, 4608)]:
        x = rnd(M, K, seed=M, scale=0.5).to(DEV)
        w = rnd(2 * I, K, seed=I, scale=0.05).to(DEV)
        assert ops.gemm_swiglu_supported(x, w, I), (M, K, I)
        gu_ref = ops.gemm(x, w)
        act_ref = ops.swiglu_fwd(gu_ref, I)
        gu, act = ops.gemm_swiglu(x, w, I)
        assert torch.equal(gu, gu_ref), (M, K, I, float((gu.float() - gu_ref.float()).abs().max()))
        assert torch.equal(act, act_ref), (M, K, I)
        # and against the oracle arithmetic (bf16 tolerances of the GEMM test)
        ref = x.float().cpu() @ w.float().cpu().t()
        close(gu[:, :64], ref[:, :64], 2e-2, 2e-2, "fused gu gate block")
        close(gu[:, I:I + 64], ref[:, I:I + 64], 2e-2, 2e-2, "fused gu up block")
    x = rnd(512, 256, seed=1).to(DEV)
    assert not ops.gemm_swiglu_supported(x, rnd(2 * 192, 256, seed=2).to(DEV), 192)          # I % 128 != 0
    with pytest.raises(Exception):
        ops.gemm_swiglu(x, rnd(2 * 192, 256, seed=2).to(DEV), 192)


@pytest.mark.parametrize("case", [(2, 333, 8, 2, [333, 256], False), (3, 513, 4, 2, None, True), (1, 2048, 32, 8, None, False)])
def test_attn_bwd_with_fused_inverse_rope_equals_two_launches(ops, case):
    """mm355_attn_bwd_rope (d == 128 kernels; inverse RoPE of dq / dk in the epilogues: dq from the accumulator fragments j / j + 4 of a lane,
    dk from the fp32 staging rows) against mm355_attn_bwd + mm355_rope_qk(inverse): dv bit for bit (untouched); dq / dk equal up to the
    rounding of one fused multiply-add (both forms rotate the bf16-rounded gradient; the compiler may contract a different product), i.e.
    <= 1 bf16 ulp on a small fraction of the elements."""
    B, L, Hq, Hkv, seqlens, off = case
    d = 128
    qkv, q, k, v, valid = _attn_setup((B, L, Hq, Hkv, d, True, seqlens), seed=5)
    do = rnd(B * L, Hq * d, seed=6, scale=0.5)
    if valid is not None:
        do = (do.view(B, L, -1) * valid[:, :, None]).reshape(B * L, -1).contiguous()
    dev = qkv.to(DEV)
    sl = torch.tensor(seqlens, dtype=torch.int32, device=DEV) if seqlens else None
    nq, nk = Hq * d, Hkv * d
    qd, kd, vd = dev[:, :nq], dev[:, nq:nq + nk], dev[:, nq + nk:]
    o, lse = ops.attn_fwd(qd, kd, vd, B, L, Hq, Hkv, d, d ** -0.5, True, sl)
    cos, sin = ops.rope_table(L + 32, d, 500000.0, DEV)
    po = torch.tensor([3, 0, 29][:B], dtype=torch.int32, device=DEV) if off else None
    ref = torch.full_like(dev, float("nan"))
    ops.attn_bwd(qd, kd, vd, o, do.to(DEV), lse, B, L, Hq, Hkv, d, d ** -0.5, True, sl, ref[:, :nq], ref[:, nq:nq + nk], ref[:, nq + nk:])
    ops.rope_qk_(ref, B, L, Hq, Hkv, d, cos, sin, inverse=True, pos_offset=po)
    got = torch.full_like(dev, float("nan"))
    ops.attn_bwd(qd, kd, vd, o, do.to(DEV), lse, B, L, Hq, Hkv, d, d ** -0.5, True, sl, got[:, :nq], got[:, nq:nq + nk], got[:, nq + nk:],
                 rope=(cos, sin, po))
    assert bool(torch.isfinite(got.float()).all())
    assert torch.equal(got[:, nq + nk:], ref[:, nq + nk:])
    for name, a, b in (("dq", got[:, :nq], ref[:, :nq]), ("dk", got[:, nq:nq + nk], ref[:, nq:nq + nk])):
        if not torch.equal(a, b):
            diff = (a.float() - b.float()).abs()
            frac = float((diff > 0).float().mean())
            rel_ulp = float((diff / b.float().abs().clamp_min(1e-4)).max())
            print(f"   {name}: {frac:.2e} of the elements differ, max relative difference {rel_ulp:.2e}")
            assert frac < 2e-2 and rel_ulp <= 2.0 ** -6, (name, case, frac, rel_ulp)


def test_gemm_rope_fused_equals_two_launches(ops):
    """mm355_gemm_rope_bf16 (q|k|v projection with the rotate-half RoPE of the q and k blocks in the epilogue; B tile staged from permuted
    weight rows so that d and d + 64 meet in one lane) writes bit for bit what mm355_gemm_bf16 + mm355_rope_qk[_pos] write: LLaMA-3-8B head
    geometry (32 / 8 heads of 128), ragged row count, with and without per-sample position offsets; and 70B's 64 / 8."""
    d = 128
    for (B, L, Hq, Hkv, K, off) in [(2, 2048, 32, 8, 4096, False), (3, 811, 32, 8, 2048, True), (2, 1024, 64, 8, 1024, False)]:
        N = (Hq + 2 * Hkv) * d
        x = rnd(B * L, K, seed=L, scale=0.5).to(DEV)
        w = rnd(N, K, seed=N, scale=0.05).to(DEV)
        cos, sin = ops.rope_table(2 * L, d, 500000.0, DEV)
        po = torch.tensor([5, 0, 17][:B], dtype=torch.int32, device=DEV) if off else None
        assert ops.gemm_rope_supported(x, w, Hq, Hkv, d, cos), (B, L, Hq, Hkv, K)
        ref = ops.gemm(x, w)
        ops.rope_qk_(ref, B, L, Hq, Hkv, d, cos, sin, pos_offset=po)
        got = ops.gemm_rope(x, w, B, L, Hq, Hkv, d, cos, sin, pos_offset=po)
        assert torch.equal(got, ref), (B, L, Hq, Hkv, K, float((got.float() - ref.float()).abs().max()))
    assert not ops.gemm_rope_supported(rnd(512, 256, seed=1).to(DEV), rnd(4 * 64, 256, seed=2).to(DEV), 2, 1, 64, cos)      # d = 64


def test_gemm_swiglu_bwd_fused_equals_two_launches(ops):
    """mm355_gemm_swiglu_bwd_bf16 (down_proj input-gradient GEMM with the SwiGLU backward in its epilogue: d act never reaches memory) against
    mm355_gemm_bf16 + mm355_swiglu_bwd_t on the same operands: dgu, actT and dguT bit for bit (the epilogue rounds d act to bf16 and uses
    the same expressions), LLaMA-3-8B widths at 4096 rows and TinyLlama's I = 5632 with a row count that is not a multiple of 256."""
    for (M, K, I) in [(4096, 4096, 14336), (2304 + 64, 2048, 5632)]:
        dy = rnd(M, K, seed=M + 1, scale=0.5).to(DEV)
        wdT = rnd(I, K, seed=I + 1, scale=0.05).to(DEV)
        gu = rnd(M, 2 * I, seed=7, scale=1.5).to(DEV)
        assert ops.gemm_swiglu_bwd_supported(dy, wdT, gu, I), (M, K, I)
        dact = ops.gemm(dy, wdT)
        dgu_ref, actT_ref, dguT_ref = ops.swiglu_bwd_t(gu, dact, I)
        dgu, actT, dguT = ops.gemm_swiglu_bwd(dy, wdT, gu, I)
        for name, got, ref in (("dgu", dgu, dgu_ref), ("actT", actT, actT_ref), ("dguT", dguT, dguT_ref)):
            same = torch.equal(got, ref)
            if not same:                                         # fp contraction may differ between the two kernels: <= 1 bf16 ulp, rare
                d = (got.float() - ref.float()).abs()
                frac = float((d > 0).float().mean())
                assert frac < 1e-3 and float((d / ref.float().abs().clamp_min(1e-6)).max()) <= 2.0 ** -7, (name, M, K, I, frac)
        assert torch.equal(dguT[:I], dgu[:, :I].t()) and torch.equal(dguT[I:], dgu[:, I:].t())     # the transposed copies ARE transposes
        act = ops.swiglu_fwd(gu, I)
        assert torch.equal(actT, act.t())
    assert not ops.gemm_swiglu_bwd_supported(rnd(512, 256, seed=1).to(DEV), rnd(192, 256, seed=2).to(DEV), rnd(512, 384, seed=3).to(DEV), 192)


# ------------------------------------------------------------------------------------------------ elementwise

def test_swiglu_gelu(ops):
    M, I = 33, 512
    gu, da = rnd(M, 2 * I, seed=1), rnd(M, I, seed=2)
    act = ops.swiglu_fwd(gu.to(DEV), I)
    g, u = gu[:, :I].float().requires_grad_(True), gu[:, I:].float().requires_grad_(True)
    ref = R.swiglu(g, u)
    close(act, ref, 1e-2, 1e-2, "swiglu fwd")
    ref.backward(da.float())
    dgu, act2 = ops.swiglu_bwd(gu.to(DEV), da.to(DEV), I)
    close(dgu[:, :I], g.grad, 1e-2, 1e-2, "swiglu dgate")
    close(dgu[:, I:], u.grad, 1e-2, 1e-2, "swiglu dup")
    assert torch.equal(act2.cpu(), act.cpu())
    x, dy = rnd(40, 256, seed=3), rnd(40, 256, seed=4)
    for kind, fn in ((0, R.gelu_erf), (1, R.gelu_tanh)):
        xf = x.float().requires_grad_(True)
        y = fn(xf)
        close(ops.gelu_fwd(x.to(DEV), kind), y, 8e-3, 8e-3, f"gelu {kind}")
        y.backward(dy.float())
        close(ops.gelu_bwd(x.to(DEV), dy.to(DEV), kind), xf.grad, 1e-2, 1e-2, f"gelu bwd {kind}")


def test_scale_axpy_cast(ops):
    x = rnd(1003, seed=1)
    s = torch.tensor([0.37], device=DEV)
    y = ops.scale_(x.to(DEV).clone(), s, 2.0)
    close(y, x.float() * 0.74, 8e-3, 1e-3, "scale")
    a, b = rnd(777, seed=2), rnd(777, seed=3)
    y = ops.axpy_(a.to(DEV).clone(), b.to(DEV), s, 1.0, True)
    close(y, a.float() + 0.37 * b.float(), 8e-3, 1e-2, "axpy")
    f = torch.randn(777, generator=torch.Generator().manual_seed(4))
    y = ops.axpy_(a.to(DEV).clone(), f.to(DEV), None, 0.5, False)
    close(y, 0.5 * f, 8e-3, 1e-3, "axpy f32")
    src = torch.randn(20, 64, generator=torch.Generator().manual_seed(5))
    dst = torch.zeros(20, 128, device=DEV, dtype=torch.bfloat16)
    ops.cast_f32_to_bf16_2d(src.to(DEV), dst[:, 32:96])
    assert torch.equal(dst[:, 32:96].cpu(), src.bfloat16())


# ------------------------------------------------------------------------------------------------ losses

def test_ce_rows(ops):
    Rr, V, ld = 37, 1003, 1024
    lg = torch.zeros(Rr, ld, dtype=torch.bfloat16)
    lg[:, :V] = rnd(Rr, V, seed=1, scale=3.0)
    lg[:, V:] = float("nan")                                    # padding must never be read arithmetically
    tg = torch.randint(0, V, (Rr,), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    tg[5] = -100
    tg[6] = V - 1
    x = lg[:, :V].float().requires_grad_(True)
    keep = tg >= 0
    lse = torch.logsumexp(x, -1)
    loss = (lse - x.gather(1, tg.clamp_min(0).long()[:, None])[:, 0])[keep].sum()
    loss.backward()
    dev = lg.to(DEV).clone()
    ls = torch.zeros(1, device=DEV)
    ops.ce_rows_(dev, tg.to(DEV), V, 0.25, ls)
    close(ls[0], loss, 2e-3, 1e-2, "ce loss sum")
    close(dev[:, :V], 0.25 * x.grad, 1e-2, 2e-4, "ce grad")
    assert float(dev[:, V:].float().abs().max()) == 0
    assert float(dev[5].float().abs().max()) == 0


@pytest.mark.parametrize("case", [dict(n=300, M=420, V=1003, h=256, gather=True), dict(n=64, M=64, V=128258, h=256, gather=False),
                                  dict(n=9001, M=9500, V=520, h=128, gather=True)])
def test_linear_ce_entry_point_through_ctypes_only(case):
    """mm355_linear_ce (SURVEY 8(b) `linear_ce`; reference metamorph_llama.py:393-413) driven with NOTHING but ctypes + raw device pointers
    (torch only owns the memory) against oracle.ref_ops: mean NLL, d hidden, d W.  n = 9001: two 8192-row chunks accumulating into an
    fp32 dW, ragged second chunk.  Two identical calls return bit-identical loss / gradients (fixed-order row sum, no atomics)."""
    import ctypes
    from metamorph_amd import lib as mmlib
    L = mmlib.load()
    n, M, V, h = case["n"], case["M"], case["V"], case["h"]
    g = torch.Generator().manual_seed(n)
    hid = (torch.randn(M, h, generator=g) * 1.0).bfloat16()
    W = (torch.randn(V, h, generator=g) * 0.08).bfloat16()
    rows = torch.randperm(M, generator=g)[:n].sort().values.to(torch.int32) if case["gather"] else None
    tgt = torch.randint(0, V, (n,), generator=g, dtype=torch.int32)
    tgt[0], tgt[-1] = V - 1, 0
    # oracle: the reference's arithmetic -- bf16 linear, .float(), mean CE (fp32) -- and autograd for the gradients
    x = (hid[rows.long()] if rows is not None else hid[:n]).float().requires_grad_(True)
    Wf = W.float().requires_grad_(True)
    logits = (x @ Wf.t()).bfloat16().float()
    logits_g = (x @ Wf.t())
    logits_g.data = logits                                          # straight-through the bf16 rounding of the logits
    loss_ref = torch.nn.functional.cross_entropy(logits_g, tgt.long())
    loss_ref.backward()
    dw_f32 = n > 8192
    d = lambda t: t.to(DEV)
    hid_d, W_d, tgt_d = d(hid), d(W), d(tgt)
    rows_d = d(rows) if rows is not None else None
    nbytes = L.mm355_linear_ce_ws_bytes(n, V, h, int(rows is not None), 1, 1)
    assert nbytes > 0
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        ws = torch.empty(nbytes, device=DEV, dtype=torch.uint8)
        loss = torch.full((1,), float("nan"), device=DEV)
        dh = torch.full((n, h), float("nan"), device=DEV, dtype=torch.bfloat16)
        dw = torch.full((V, h), float("nan"), device=DEV, dtype=torch.float32 if dw_f32 else torch.bfloat16)
        rc = L.mm355_linear_ce(hid_d.data_ptr(), h, rows_d.data_ptr() if rows_d is not None else None, tgt_d.data_ptr(), n, W_d.data_ptr(), h, V, h,
                               loss.data_ptr(), dh.data_ptr(), dw.data_ptr(), int(dw_f32), ws.data_ptr(), nbytes, stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return loss, dh, dw

    loss, dh, dw = run()
    close(loss[0], loss_ref.detach(), 1e-3, 1e-4, "linear_ce loss")
    close(dh, x.grad, 2e-2, 2e-2 * float(x.grad.abs().max()), "linear_ce d hidden")
    close(dw, Wf.grad, 2e-2, 2e-2 * float(Wf.grad.abs().max()), "linear_ce d W")
    loss2, dh2, dw2 = run()
    assert torch.equal(loss, loss2) and torch.equal(dh, dh2) and torch.equal(dw, dw2)          # bit-reproducible
    # loss only (evaluation): no gradient buffers, a smaller workspace
    nb0 = L.mm355_linear_ce_ws_bytes(n, V, h, int(rows is not None), 0, 0)
    assert 0 < nb0 < nbytes
    ws0, l0 = torch.empty(nb0, device=DEV, dtype=torch.uint8), torch.zeros(1, device=DEV)
    assert L.mm355_linear_ce(hid_d.data_ptr(), h, rows_d.data_ptr() if rows_d is not None else None, tgt_d.data_ptr(), n, W_d.data_ptr(), h, V, h,
                             l0.data_ptr(), None, None, 0, ws0.data_ptr(), nb0, stream) == 0
    assert torch.equal(l0, loss)
    assert L.mm355_linear_ce(hid_d.data_ptr(), h, None, tgt_d.data_ptr(), n, W_d.data_ptr(), h, V, h, l0.data_ptr(), None, None, 0, ws0.data_ptr(),
                             nb0 // 2, stream) == -1                                               # workspace too small: MM355_EINVAL


def test_loss_scalars_are_bit_reproducible(ops):
    """CE / cosine / soft-CE / mean-abs sums go through per-row values + ONE fixed-order reduction (row_ws), not one fp32 atomicAdd per row:
    five calls on the same inputs return the same bits (DESIGN section 8 carried "loss scalar not bit-reproducible" since round 2)."""
    Rr, C, V = 4099, 1152, 2048
    p, t = rnd(Rr, C, seed=11).to(DEV), rnd(Rr, C, seed=12).to(DEV)
    lg = rnd(Rr, V, seed=13, scale=3.0).to(DEV)
    tg = torch.randint(0, V, (Rr,), generator=torch.Generator().manual_seed(14), dtype=torch.int32).to(DEV)

    def once():
        ls = torch.zeros(1, device=DEV)
        ops.ce_rows_(lg.clone(), tg, V, 1.0, ls)
        return (ls.clone(), ops.cosine_loss(p, t, 1)[0].clone(), ops.soft_ce_loss(p, ops.softmax_rows(t), True)[0].clone(),
                ops.mean_abs_loss(p, t)[0].clone())

    first = once()
    for _ in range(4):
        assert all(torch.equal(a, b) for a, b in zip(first, once()))
    # the atomics form (row_ws = NULL, kept in the ABI) sums the same values: equal up to summation order
    ls_a = torch.zeros(1, device=DEV)
    ops.ce_rows_(lg.clone(), tg, V, 1.0, ls_a, deterministic=False)
    close(ls_a, first[0], 1e-5, 0, "atomics vs fixed order")


def test_gemm_small_m_tiles_are_bit_identical_and_selected_for_prompt_shapes(ops):
    """Round 6: 64 x 128 tiles (variant 9) for prompt-pass shapes -- a few hundred rows against N = 4096 .. 6144, where 128 x 128 tiles leave
    half of the CUs without a workgroup.  Same K order per output element as every other tiling: bit-identical to variants 2 and 11, with
    every epilogue flag; and the auto dispatch (variant 0) returns those same bits on such shapes."""
    for (M, N, K) in [(512, 4096, 4096), (520, 6144, 4096), (300, 4096, 14336), (729, 1152, 1152), (96, 1024, 256)]:
        a, b = rnd(M, K, seed=M).to(DEV), rnd(N, K, seed=N, scale=0.05).to(DEV)
        ref = ops.gemm(a, b, variant=2)
        assert torch.equal(ops.gemm(a, b, variant=9), ref) and torch.equal(ops.gemm(a, b, variant=0), ref), (M, N, K)
        if K % 128 == 0:
            assert torch.equal(ops.gemm(a, b, variant=11), ref)
        bias, res = rnd(N, seed=3).to(DEV), rnd(M, N, seed=4).to(DEV)
        for kw in (dict(bias=bias, gelu="tanh"), dict(residual=res), dict(out_f32=True), dict(bias=bias, residual=res, gelu="erf")):
            assert torch.equal(ops.gemm(a, b, variant=9, **kw), ops.gemm(a, b, variant=2, **kw)), (M, N, K, sorted(kw))
        acc9, acc2 = res.clone(), res.clone()
        ops.gemm(a, b, out=acc9, accumulate=True, variant=9)
        ops.gemm(a, b, out=acc2, accumulate=True, variant=2)
        assert torch.equal(acc9, acc2)


@pytest.mark.parametrize("shape", [(512, 4096, 14336), (128, 6144, 4096), (520, 4096, 4096), (1000, 1152, 4352), (64, 1024, 512), (2048, 28672, 4096)])
def test_gemm_splitk_against_the_plain_kernel_and_fp64(ops, shape):
    """mm355_gemm_splitk_bf16 (the prompt-pass form: K slices as separate workgroups of one launch, fp32 partials summed in slice order):
    against an fp64 product within one bf16 rounding + the fp32 accumulation bound, against mm355_gemm_bf16 within one bf16 step (another
    summation order), with and without the residual; shapes it does not split ((64, 1024, 512): K too short; (2048, 28672, 4096): enough
    tiles) forward to the plain kernel and ARE bit-identical; two runs of the split form are bit-identical (fixed slice order)."""
    from metamorph_amd import lib as mmlib
    M, N, K = shape
    a, b, res = rnd(M, K, seed=M), rnd(N, K, seed=N, scale=0.05), rnd(M, N, seed=7)
    ad, bd, rd = a.to(DEV), b.to(DEV), res.to(DEV)
    split = int(mmlib.load().mm355_gemm_splitk_ws_floats(M, N, K)) > 0
    assert split == (shape not in ((64, 1024, 512), (2048, 28672, 4096)))
    ref = a.double() @ b.double().t()
    bound = 4.0 * K * 2.0 ** -24 * (a.double().abs() @ b.double().abs().t())
    for r, rr in ((None, 0.0), (rd, res.double())):
        got = ops.gemm_splitk(ad, bd, residual=r)
        plain = ops.gemm(ad, bd, residual=r)
        want = ref + rr
        err = (got.double().cpu() - want).abs()
        assert bool((err <= 2.0 ** -8 * want.abs() + bound).all()), (shape, r is not None, float(err.max()))
        if split:
            d = (got.float() - plain.float()).abs()
            # one bf16 step; near-zero outputs (a cancellation of K terms of size ~3: |value| ~ 1e-5) carry the fp32 summation noise itself
            assert bool((d <= 2.0 ** -7 * torch.maximum(plain.float().abs(), got.float().abs()) + 2e-4).all()), (shape, float(d.max()))
            assert torch.equal(got, ops.gemm_splitk(ad, bd, residual=r))
        else:
            assert torch.equal(got, plain)


@pytest.mark.parametrize("M", [17, 32])
def test_gemm_splitk_32_row_tiles_give_the_bits_of_64_row_tiles(ops, M):
    """Up to 32 rows the split-K launch takes 32 x 128 tiles instead of 64 x 128 (the decode step of 17 - 32 sequences): every output element
    sums its K slice in the same order under either tile -- the rows of an M-row call equal the same rows inside a 64-row call (64-row tiles,
    same number of slices) bit for bit, on all four LLaMA-3-8B projections."""
    for (N, K) in ((6144, 4096), (4096, 4096), (4096, 14336), (28672, 4096)):
        a, b = rnd(64, K, seed=M + N).to(DEV), rnd(N, K, seed=K, scale=0.03).to(DEV)
        assert torch.equal(ops.gemm_splitk(a[:M], b), ops.gemm_splitk(a, b)[:M]), (M, N, K)


def test_cosine_loss(ops):
    Rr, C = 21, 1152
    p, t = rnd(Rr, C, seed=1), R.l2_normalize(rnd(Rr, C, seed=2).float()).bfloat16()
    for normalize in (1, 0):
        pf = p.float().requires_grad_(True)
        u = R.l2_normalize(pf) if normalize else pf
        loss = R.cosine_loss(t.float(), u)
        loss.backward()
        cs, dp = ops.cosine_loss(p.to(DEV), t.to(DEV), normalize)
        close(-cs[0] / Rr, loss, 1e-2, 2e-3, f"cosine loss normalize={normalize}")
        close(dp, pf.grad, 3e-2, 2e-2 * float(pf.grad.abs().max()), f"cosine grad normalize={normalize}")


def test_mean_abs_loss(ops):
    """The reference's `mse_loss_fn` (mean |z - h|, metamorph_llama.py:211-219): value and gradient against the oracle."""
    Rr, C = 37, 1152
    p, t = rnd(Rr, C, seed=3), rnd(Rr, C, seed=4)
    t[5, :16] = p[5, :16]                                        # exact ties: sign(0) = 0 like torch.abs backward
    pf = p.float().requires_grad_(True)
    loss = R.mean_abs_loss(t.float(), pf)
    loss.backward()
    s, dp = ops.mean_abs_loss(p.to(DEV), t.to(DEV))
    close(s[0] / (Rr * C), loss, 2e-3, 1e-5, "mean-abs loss")     # bf16 difference (reference stack) vs fp32 difference
    close(dp, pf.grad, 1e-2, 0.0, "mean-abs grad")
    assert float(dp[5, :16].float().abs().max()) == 0


@pytest.mark.parametrize("normalize", [True, False])
def test_soft_ce_loss(ops, normalize):
    """apply_softmax head (metamorph_llama.py:434-447): -(t * log(softmax(u / 0.07) + 1e-10)).sum(1).mean() with soft targets."""
    Rr, C = 19, 1152
    scale = 1.0 if normalize else 0.03                           # raw predictions / 0.07 must not saturate the softmax
    p = rnd(Rr, C, seed=5, scale=scale)
    t = torch.softmax(R.l2_normalize(rnd(Rr, C, seed=6).float()) / 0.07, -1).bfloat16()
    pf = p.float().requires_grad_(True)
    u = R.l2_normalize(pf) if normalize else pf
    loss = R.soft_ce_loss(t.float(), torch.softmax(u / 0.07, -1))
    loss.backward()
    s, dp = ops.soft_ce_loss(p.to(DEV), t.to(DEV), normalize)
    close(s[0] / Rr, loss, 5e-3, 1e-3, f"soft-CE loss normalize={normalize}")
    close(dp, pf.grad, 5e-2, 3e-2 * float(pf.grad.abs().max()), f"soft-CE grad normalize={normalize}")


def test_softmax_rows(ops):
    """softmax(x / 0.07) rows (siglip_encoder.py:210-211, metamorph_llama.py:372-373) and its backward."""
    Rr, C = 23, 1152
    x = R.l2_normalize(rnd(Rr, C, seed=7).float()).bfloat16()
    xf = x.float().requires_grad_(True)
    y = torch.softmax(xf / 0.07, -1)
    dy = rnd(Rr, C, seed=8)
    (y * dy.float()).sum().backward()
    got = ops.softmax_rows(x.to(DEV), 0.07)
    close(got, y, 2e-2, 1e-6, "softmax rows")
    assert abs(float(got.float().sum(-1).mean()) - 1.0) < 5e-3
    dx = ops.softmax_rows_bwd(got, dy.to(DEV), 0.07)
    close(dx, xf.grad, 5e-2, 3e-2 * float(xf.grad.abs().max()), "softmax rows bwd")


# ------------------------------------------------------------------------------------------------ splice

def test_splice_and_rows(ops):
    V, h = 500, 256
    emb, proj = rnd(V, h, seed=1), rnd(12, h, seed=2)
    src = torch.tensor([3, 499, -1, -2, -13, 0, 3, -1, -7], dtype=torch.int32)
    out = ops.splice_gather(emb.to(DEV), proj.to(DEV), src.to(DEV), h)
    for r, s in enumerate(src.tolist()):
        exp = emb[s] if s >= 0 else (torch.zeros(h, dtype=torch.bfloat16) if s == -1 else proj[-2 - s])
        assert torch.equal(out[r].cpu(), exp), r
    x = rnd(40, h, seed=3)
    idx = torch.tensor([5, -1, 39, 0], dtype=torch.int32)
    g = ops.rows_gather(x.to(DEV), idx.to(DEV))
    assert torch.equal(g[0].cpu(), x[5]) and torch.equal(g[2].cpu(), x[39]) and float(g[1].abs().max()) == 0
    dst = x.to(DEV).clone()
    add = rnd(4, h, seed=4)
    ops.rows_scatter_add_(dst, add.to(DEV), idx.to(DEV))
    ref = x.float().clone()
    for r, i in enumerate(idx.tolist()):
        if i >= 0:
            ref[i] += add[r].float()
    close(dst, ref, 8e-3, 1e-2, "rows_scatter_add")
    # embedding gradient by sorted segments
    dout = rnd(9, h, seed=5)
    tok = torch.tensor([0, 3, 499], dtype=torch.int32)
    seg = torch.tensor([0, 1, 3, 4], dtype=torch.int32)
    pos = torch.tensor([5, 0, 6, 1], dtype=torch.int32)
    de = torch.zeros(V, h, device=DEV, dtype=torch.bfloat16)
    ops.embed_grad_(de, dout.to(DEV), tok.to(DEV), seg.to(DEV), pos.to(DEV), False)
    close(de[3], dout[0].float() + dout[6].float(), 8e-3, 1e-2, "embed grad dup token")
    assert torch.equal(de[0].cpu(), dout[5]) and torch.equal(de[499].cpu(), dout[1])
    ops.embed_grad_(de, dout.to(DEV), tok.to(DEV), seg.to(DEV), pos.to(DEV), True)
    close(de[3], 2 * (dout[0].float() + dout[6].float()), 1e-2, 2e-2, "embed grad accumulate")


def test_swiglu_bwd_transposed_outputs(ops):
    """mm355_swiglu_bwd_t == mm355_swiglu_bwd + transposes of act and dgu, bit for bit."""
    M, I = 192, 320
    gu, dact = rnd(M, 2 * I, seed=1).to(DEV), rnd(M, I, seed=2).to(DEV)
    dgu0, act0 = ops.swiglu_bwd(gu, dact, I)
    dgu, actT, dguT = ops.swiglu_bwd_t(gu, dact, I)
    assert torch.equal(dgu, dgu0) and torch.equal(actT, act0.t()) and torch.equal(dguT, dgu0.t())
    from metamorph_amd.lib import Mm355Error
    with pytest.raises(Mm355Error):
        ops.swiglu_bwd_t(rnd(100, 2 * I, seed=3).to(DEV), rnd(100, I, seed=4).to(DEV), I)


# ------------------------------------------------------------------------------------------------ decode shape (row N1)

@pytest.mark.parametrize("M", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("NK", [(64, 512), (130, 1032), (6144, 4096), (1000, 14336), (4096, 64), (40, 16896)])
def test_gemv(ops, M, NK):
    """(K = 14336 / 16896: the x rows pass through LDS in windows of 4096 columns)"""
    N, K = NK
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = x.float() @ w.float().t()
    close(ops.gemv(x.to(DEV), w.to(DEV)), ref, 1e-2, 0.02, f"gemv {M}x{N}x{K}")
    close(ops.gemv(x.to(DEV), w.to(DEV), bias=b.to(DEV), gelu="erf"), R.gelu_erf(ref + b.float()), 1e-2, 0.02, "gemv bias+gelu")
    close(ops.gemv(x.to(DEV), w.to(DEV), residual=r.to(DEV)), ref + r.float(), 1e-2, 0.03, "gemv residual")
    of = torch.empty(M, N, device=DEV, dtype=torch.float32)
    close(ops.gemv(x.to(DEV), w.to(DEV), out=of), ref, 1e-4, 2e-3, "gemv f32")
    wide = rnd(N, K + 64, seed=5, scale=0.05).to(DEV)            # strided weight rows
    close(ops.gemv(x.to(DEV), wide[:, 32:32 + K]), x.float() @ wide[:, 32:32 + K].float().cpu().t(), 1e-2, 0.02, "gemv strided")


def test_gemv_rejects_many_rows(ops):
    from metamorph_amd.lib import Mm355Error
    with pytest.raises(Mm355Error):
        ops.gemv(rnd(17, 64, seed=1).to(DEV), rnd(16, 64, seed=2).to(DEV))


@pytest.mark.parametrize("M", [17, 32, 64, 130])
def test_gemm_splitk_fused_reduce_launches_equal_the_launch_sequences(ops, M):
    """mm355_gemm_splitk_{norm,swiglu,rope_append}_bf16 (what follows a split projection folded into its reduce launch) against
    gemm_splitk -> rmsnorm_fwd / swiglu_fwd / rope_kv_append_: bit for bit, on shapes that are split (LLaMA-3-8B widths: K = 4096 / 14336)
    and on shapes that are not (few K tiles: the library runs the plain sequence itself)."""
    for (N, K) in ((4096, 14336), (4096, 4096), (512, 256)):
        a, b, r = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=0.03).to(DEV), rnd(M, N, seed=3).to(DEV)
        nw = (1.0 + 0.1 * rnd(N, seed=4)).bfloat16().to(DEV)
        c0 = ops.gemm_splitk(a, b, residual=r)
        y0 = ops.rmsnorm_fwd(c0, nw, 1e-5)
        c1, y1 = ops.gemm_splitk_norm(a, b, nw, 1e-5, residual=r)
        assert torch.equal(c0, c1) and torch.equal(y0, y1), ("norm", N, K)
        c2, y2 = ops.gemm_splitk_norm(a, b, nw, 1e-5)
        assert torch.equal(c2, ops.gemm_splitk(a, b)) and torch.equal(y2, ops.rmsnorm_fwd(c2, nw, 1e-5)), ("norm, no residual", N, K)
    for (I, K) in ((14336, 4096), (256, 128)):
        x, w = rnd(M, K, seed=5).to(DEV), rnd(2 * I, K, seed=6, scale=0.05).to(DEV)
        assert torch.equal(ops.gemm_splitk_swiglu(x, w, I), ops.swiglu_fwd(ops.gemm_splitk(x, w), I)), ("swiglu", I, K)
    for (Hq, Hkv, d, K) in ((32, 8, 128, 4096), (4, 2, 64, 128)):
        if M > 64 and K == 4096:
            continue                                                    # (one row per sequence: decode batches)
        N, Lmax = (Hq + 2 * Hkv) * d, 50
        x, w = rnd(M, K, seed=7).to(DEV), rnd(N, K, seed=8, scale=0.05).to(DEV)
        cos, sin = ops.rope_table(Lmax, d, 10000.0, DEV)
        pos = torch.tensor([(7 * m + 3) % Lmax for m in range(M)], dtype=torch.int32, device=DEV)
        k0, v0 = rnd(M, Lmax, Hkv * d, seed=9).to(DEV), rnd(M, Lmax, Hkv * d, seed=10).to(DEV)
        k1, v1 = k0.clone(), v0.clone()
        qkv = ops.gemm_splitk(x, w)
        ops.rope_kv_append_(qkv, Hq, Hkv, d, cos, sin, pos, k0, v0)
        got = ops.gemm_splitk_rope_append(x, w, Hq, Hkv, d, cos, sin, pos, k1, v1)
        assert torch.equal(got[:, :Hq * d], qkv[:, :Hq * d]), ("q rows", Hq, d)
        assert torch.equal(k1, k0) and torch.equal(v1, v0), ("cache rows", Hq, d)


@pytest.mark.parametrize("M", [17, 24, 32])
def test_gemv_second_row_group_on_wide_weights(ops, M):
    """17 .. 32 rows on the MFMA GEMVs (round 6): on weights wide enough that every workgroup owns one K range (gate|up, lm_head) a second
    group of 16 x rows rides on the same weight fragments -- the weights are streamed once.  Every row equals the 16-row call on its slice bit
    for bit (plain, bias + GELU, residual, fp32 output; the SwiGLU epilogue against gemv -> swiglu_fwd); narrower weights are refused (the
    callers take the split-K GEMM there)."""
    from metamorph_amd.lib import Mm355Error
    N, K = 20992, 1032                                                      # 1312 groups of 16 weight rows; K with a tail step
    x, w, b, r = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=0.05).to(DEV), rnd(N, seed=3).to(DEV), rnd(M, N, seed=4).to(DEV)
    sl = (slice(0, 16), slice(16, M))
    pad16 = lambda t: t if t.shape[0] == 16 else torch.cat([t, t.new_zeros(16 - t.shape[0], t.shape[1])], 0)      # (every slice as a 16-row call: the MFMA form)
    rows16 = lambda f: torch.cat([f(pad16(x[c]), pad16(r[c]))[:x[c].shape[0]] for c in sl], 0)
    got = ops.gemv(x, w)
    close(got, x.float().cpu() @ w.float().cpu().t(), 1e-2, 0.02, f"gemv {M}x{N}x{K}")
    assert torch.equal(got, rows16(lambda xx, rr: ops.gemv(xx, w)))
    assert torch.equal(ops.gemv(x, w, bias=b, gelu="erf"), rows16(lambda xx, rr: ops.gemv(xx, w, bias=b, gelu="erf")))
    assert torch.equal(ops.gemv(x, w, residual=r), rows16(lambda xx, rr: ops.gemv(xx, w, residual=rr)))
    of = torch.empty(M, N, device=DEV, dtype=torch.float32)
    assert torch.equal(ops.gemv(x, w, out=of), rows16(lambda xx, rr: ops.gemv(xx, w, out=torch.empty(16, N, device=DEV, dtype=torch.float32))))
    I, K2 = 10496, 4096                                                     # I / 8 = 1312 groups
    x2, w2 = rnd(M, K2, seed=5).to(DEV), rnd(2 * I, K2, seed=6, scale=0.05).to(DEV)
    assert torch.equal(ops.gemv_swiglu(x2, w2, I), torch.cat([ops.gemv_swiglu(pad16(x2[c]), w2, I)[:x2[c].shape[0]] for c in sl], 0))
    assert torch.equal(ops.gemv_swiglu(x2, w2, I), ops.swiglu_fwd(torch.cat([ops.gemv(pad16(x2[c]), w2)[:x2[c].shape[0]] for c in sl], 0), I))
    with pytest.raises(Mm355Error):
        ops.gemv(x, w[:4096])                                               # 256 groups: K is split over the waves of a workgroup there
    with pytest.raises(Mm355Error):
        ops.gemv_swiglu(x2, w2, I, norm_w=(1.0 + 0.1 * rnd(K2, seed=7)).bfloat16().to(DEV), eps=1e-5)   # the folded norm keeps 16 rows


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("case", [(1, 8, 2, 128, [700]), (3, 4, 4, 64, [1, 256, 300]), (2, 32, 4, 128, [513, 77]), (1, 16, 16, 72, [40]),
                                  (2, 8, 2, 128, [2500, 1030]), (2, 16, 2, 128, [1024, 1025]), (1, 4, 2, 64, [4000]), (3, 32, 8, 128, [128, 129, 127])])
def test_attn_decode(ops, case, variant):
    """Last-row attention against a KV cache == the oracle's attention on the same prefix, last query row.  Variant 0 (product): one
    1024-thread workgroup per 1024 cached rows -- caches of <= 1024 rows finish in it, longer ones through one record per group merged
    by the group that arrives last; variant 1: 256-row chunks + merge by the last (round 3's split, one launch)."""
    B, Hq, Hkv, d, lens = case
    Lmax = max(lens) + 5
    g = torch.Generator().manual_seed(7)
    q = (torch.randn(B, Hq * d, generator=g) * 0.7).bfloat16()
    kc = (torch.randn(B, Lmax, Hkv * d, generator=g) * 0.7).bfloat16()
    vc = (torch.randn(B, Lmax, Hkv * d, generator=g) * 0.7).bfloat16()
    out = ops.attn_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV), max(lens), Hq, Hkv, d, d ** -0.5,
                          variant=variant)
    for b in range(B):
        n = lens[b]
        qq = q[b].view(Hq, 1, d).float()
        kk = kc[b, :n].view(n, Hkv, d).transpose(0, 1).float().repeat_interleave(Hq // Hkv, dim=0)      # [Hq, n, d]
        vv = vc[b, :n].view(n, Hkv, d).transpose(0, 1).float().repeat_interleave(Hq // Hkv, dim=0)
        ref = (torch.softmax(qq @ kk.transpose(1, 2) * d ** -0.5, dim=-1) @ vv).reshape(Hq * d)
        close(out[b], ref, 1e-2, 1e-2, f"attn_decode {case} sample {b}")


def test_attn_decode_takes_a_q_view_without_16_byte_alignment(ops):
    """The kernel loads a lane's q fragment with one 16-B load where q allows it and with 2-byte loads otherwise (q carries no alignment
    contract in include/mm355.h): a q view that starts 2 bytes into its rows gives the bits of the aligned copy."""
    B, Hq, Hkv, d = 3, 32, 8, 128
    g = torch.Generator().manual_seed(5)
    wide = (torch.randn(B, Hq * d + 8, generator=g) * 0.7).bfloat16().to(DEV)
    q_off = wide[:, 1:1 + Hq * d]
    assert q_off.data_ptr() % 16 != 0
    kc = (torch.randn(B, 2048, Hkv * d, generator=g) * 0.7).bfloat16().to(DEV)
    vc = (torch.randn(B, 2048, Hkv * d, generator=g) * 0.7).bfloat16().to(DEV)
    for lens, bound in (([700, 64, 1], 1024), ([1500, 700, 1], 2048)):      # one key group (one head per workgroup) / two (the GQA group per workgroup)
        kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
        a = ops.attn_decode(q_off, kc, vc, kv, bound, Hq, Hkv, d, d ** -0.5)
        b = ops.attn_decode(q_off.contiguous(), kc, vc, kv, bound, Hq, Hkv, d, d ** -0.5)
        assert torch.equal(a, b), bound


@pytest.mark.parametrize("shape", [(1, 32, 8, 128), (8, 32, 8, 128), (3, 16, 8, 128), (2, 32, 4, 64), (16, 32, 8, 128)])
def test_attn_decode_heads_spread_over_workgroups_is_bit_identical(ops, shape):
    """With a bound of <= 1024 cached rows the product launch gives every query head (or pair of heads) of a GQA group its own workgroup;
    variant 2 keeps the group in one workgroup (the form used for longer bounds).  Each head's arithmetic is the same: equal bit for bit,
    for full, short, one-row and empty caches, with a bound of exactly 1024 and below."""
    B, Hq, Hkv, d = shape
    g = torch.Generator().manual_seed(31)
    q = (torch.randn(B, Hq * d, generator=g) * 1.5).bfloat16().to(DEV)
    kc = (torch.randn(B, 1024, Hkv * d, generator=g) * 1.5).bfloat16().to(DEV)
    vc = torch.randn(B, 1024, Hkv * d, generator=g).bfloat16().to(DEV)
    lens = [1024, 1, 577, 0, 256, 1023, 64, 300, 5, 900, 2, 1000, 129, 640, 33, 17][:B]
    kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
    for bound in (1024, max(max(lens), 1)):
        a = ops.attn_decode(q, kc, vc, kv, bound, Hq, Hkv, d, d ** -0.5, variant=0)
        b = ops.attn_decode(q, kc, vc, kv, bound, Hq, Hkv, d, d ** -0.5, variant=2)
        assert torch.equal(a, b), (shape, bound)
    # and the same rows under the capacity bound of a longer cache (whole group per workgroup, second key group idle)
    kc2 = torch.cat([kc, kc[:, :512]], 1).contiguous()
    vc2 = torch.cat([vc, vc[:, :512]], 1).contiguous()
    c = ops.attn_decode(q, kc2, vc2, kv, 1536, Hq, Hkv, d, d ** -0.5, variant=0)
    assert torch.equal(ops.attn_decode(q, kc, vc, kv, 1024, Hq, Hkv, d, d ** -0.5), c)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("kind", ["wide", "sink", "cliff"])
def test_attn_decode_on_hostile_scores(ops, kind, variant):
    """mm355_attn_decode where a trained checkpoint puts it: logits of +-30 .. 60 (q, k ~ N(0, 3^2)), an attention sink at key 0 (+40 over
    everything else), and a cliff -- every score at -40 except one key at +40 that sits in the LAST group of a long cache (the groups'
    partial maxima differ by 115 log2 units when they are merged).  The kernel scales in fp32 (no pre-rounded operand), so it must stay at
    bf16-output accuracy at any magnitude."""
    B, Hq, Hkv, d, lens = 2, 32, 8, 128, [2500, 700]
    Lmax = max(lens) + 3
    g = torch.Generator().manual_seed(11)
    sig = 3.0 if kind == "wide" else 0.5
    q = (torch.randn(B, Hq, d, generator=g) * sig).bfloat16()
    kc = (torch.randn(B, Lmax, Hkv, d, generator=g) * sig).bfloat16()
    vc = torch.randn(B, Lmax, Hkv, d, generator=g).bfloat16()
    unit = d ** -0.5 * 8.0
    if kind != "wide":
        q[..., 0] = 8.0
        kc[..., 0] = 0
    if kind == "sink":
        kc[:, 0, :, 0] = 40.0 / unit
    if kind == "cliff":
        kc[..., 0] = -40.0 / unit
        for b in range(B):
            kc[b, lens[b] - 7, :, 0] = 40.0 / unit
    out = ops.attn_decode(q.view(B, Hq * d).to(DEV), kc.view(B, Lmax, Hkv * d).to(DEV), vc.view(B, Lmax, Hkv * d).to(DEV),
                          torch.tensor(lens, dtype=torch.int32, device=DEV), max(lens), Hq, Hkv, d, d ** -0.5, variant=variant)
    for b in range(B):
        n = lens[b]
        kk = kc[b, :n].transpose(0, 1).float().repeat_interleave(Hq // Hkv, dim=0)      # [Hq, n, d]
        vv = vc[b, :n].transpose(0, 1).float().repeat_interleave(Hq // Hkv, dim=0)
        s = (q[b].float()[:, None] @ kk.transpose(1, 2)) * d ** -0.5
        ref = (torch.softmax(s, dim=-1) @ vv).reshape(Hq * d)
        print(f"   decode {kind} sample {b}: max |s| {float(s.abs().max()):.1f}, max err {float((out[b].float().cpu() - ref).abs().max()):.2e}")
        close(out[b], ref, 2.0 ** -7, 2.0 ** -7, f"attn_decode {kind} sample {b}")


def test_ce_rows_with_logits_of_80(ops):
    """mm355_ce_rows on logits of +-80 (a confident checkpoint: the target or one wrong class dominates by e^160): loss and the in-place
    softmax gradient against fp32; no overflow, exact zeros where the probability underflows."""
    Rr, V, ld = 64, 5003, 5120
    g = torch.Generator().manual_seed(3)
    lg = torch.zeros(Rr, ld, dtype=torch.bfloat16)
    base = torch.randn(Rr, V, generator=g) * 20.0
    tg = torch.randint(0, V, (Rr,), generator=g, dtype=torch.int32)
    for r in range(Rr):
        base[r, int(tg[r]) if r % 2 == 0 else (int(tg[r]) + 1) % V] = 80.0      # even rows: the target dominates; odd rows: a wrong class does
        base[r, (int(tg[r]) + 2) % V] = -80.0
    lg[:, :V] = base.bfloat16()
    lg[:, V:] = float("nan")
    tg[7] = -100
    x = lg[:, :V].float().requires_grad_(True)
    keep = tg >= 0
    loss = (torch.logsumexp(x, -1) - x.gather(1, tg.clamp_min(0).long()[:, None])[:, 0])[keep].sum()
    loss.backward()
    dev = lg.to(DEV).clone()
    ls = torch.zeros(1, device=DEV)
    ops.ce_rows_(dev, tg.to(DEV), V, 1.0, ls)
    assert bool(torch.isfinite(dev[:, :V].float()).all()) and bool(torch.isfinite(ls).all())
    close(ls[0], loss, 2e-3, 1e-2, "ce loss sum at |logit| = 80")
    close(dev[:, :V], x.grad, 1e-2, 2e-4, "ce grad at |logit| = 80")
    assert float(dev[:, V:].float().abs().max()) == 0 and float(dev[7].float().abs().max()) == 0


@pytest.mark.parametrize("M", [1, 2, 3, 5, 8, 16])
@pytest.mark.parametrize("IK", [(64, 512), (14336, 4096), (1000, 1032), (64, 8192), (48, 5120)])
def test_gemv_swiglu_fused_equals_the_launch_sequence(ops, M, IK):
    """mm355_gemv_swiglu_bf16 (RMSNorm in the operand read, SiLU(g) * u in the epilogue) == rmsnorm_fwd -> gemv -> swiglu_fwd, bit for bit;
    and without the norm == gemv -> swiglu_fwd.  (K = 5120 / 8192: rows longer than one 4096-column LDS window -- the norm's sum of squares
    then comes from a pass of its own in front of the stream, the windows are normalised as they are staged.)"""
    I, K = IK
    x, w, nw = rnd(M, K, seed=1).to(DEV), rnd(2 * I, K, seed=2, scale=0.05).to(DEV), (1.0 + 0.1 * rnd(K, seed=3)).bfloat16().to(DEV)
    ref = ops.swiglu_fwd(ops.gemv(ops.rmsnorm_fwd(x, nw, 1e-5), w), I)
    assert torch.equal(ops.gemv_swiglu(x, w, I), ops.swiglu_fwd(ops.gemv(x, w), I))
    if M > 4 and M * ((K + 31) // 32 * 32 + 8) * 2 > 140 * 1024:
        # 5 .. 16 rows with the norm folded in keep ALL normalised rows in LDS (no windows on the MFMA form): refused beyond 140 KiB,
        # the caller runs mm355_rmsnorm_fwd first (functional.py folds the norms up to four rows only)
        from metamorph_amd.lib import Mm355Error
        with pytest.raises(Mm355Error):
            ops.gemv_swiglu(x, w, I, norm_w=nw, eps=1e-5)
        return
    assert torch.equal(ops.gemv_swiglu(x, w, I, norm_w=nw, eps=1e-5), ref)


@pytest.mark.parametrize("geo", [(1, 8, 2, 128, 512), (3, 4, 4, 64, 1032), (2, 32, 8, 128, 4096), (8, 2, 1, 80, 256), (8, 32, 8, 128, 4096), (16, 8, 2, 128, 1024),
                                 (5, 8, 2, 128, 8192), (2, 4, 4, 64, 5120), (1, 8, 2, 128, 12288)])
def test_gemv_rope_append_fused_equals_the_launch_sequence(ops, geo):
    """mm355_gemv_rope_append_bf16 == rmsnorm_fwd -> gemv -> rope_kv_append: the q columns of the row buffer and the cache rows written
    (and ONLY those cache rows) carry the same bits; positions differ per sample and come from device memory."""
    B, Hq, Hkv, d, K = geo
    N, Lmax = (Hq + 2 * Hkv) * d, 40
    x, w, nw = rnd(B, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=0.05).to(DEV), (1.0 + 0.1 * rnd(K, seed=3)).bfloat16().to(DEV)
    cos, sin = ops.rope_table(Lmax, d, 10000.0, DEV)
    pos = torch.tensor([(7 * b + 3) % Lmax for b in range(B)], dtype=torch.int32, device=DEV)
    base_k, base_v = rnd(B, Lmax, Hkv * d, seed=4).to(DEV), rnd(B, Lmax, Hkv * d, seed=5).to(DEV)
    for norm in (True, False):
        k0, v0, k1, v1 = base_k.clone(), base_v.clone(), base_k.clone(), base_v.clone()
        qkv = ops.gemv(ops.rmsnorm_fwd(x, nw, 1e-5) if norm else x, w)
        ops.rope_kv_append_(qkv, Hq, Hkv, d, cos, sin, pos, k0, v0)
        got = ops.gemv_rope_append(x, w, Hq, Hkv, d, cos, sin, pos, k1, v1, norm_w=nw if norm else None, eps=1e-5)
        assert torch.equal(got[:, :Hq * d], qkv[:, :Hq * d]), "q rows"
        assert torch.equal(k1, k0) and torch.equal(v1, v0), "cache rows"


@pytest.mark.parametrize("variant,lens", [(0, [2700, 300]), (0, [700, 300]), (0, [100, 3000]), (1, [700, 300])])
def test_attn_decode_counters_return_to_zero_and_replay(ops, variant, lens):
    """The chunk that arrives last merges and re-arms its counter: the same workspace serves launch after launch (hipGraph replay)."""
    B, Hq, Hkv, d = 2, 32, 8, 128
    g = torch.Generator().manual_seed(9)
    q = (torch.randn(B, Hq * d, generator=g) * 0.7).bfloat16().to(DEV)
    kc = (torch.randn(B, 3072, Hkv * d, generator=g) * 0.7).bfloat16().to(DEV)
    vc = (torch.randn(B, 3072, Hkv * d, generator=g) * 0.7).bfloat16().to(DEV)
    kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
    ws = torch.zeros(int(ops._L().mm355_attn_decode_ws_floats(B, Hq, d, 3072)), device=DEV, dtype=torch.float32)
    first = ops.attn_decode(q, kc, vc, kv, 3072, Hq, Hkv, d, d ** -0.5, workspace=ws, variant=variant).clone()
    for _ in range(5):
        assert torch.equal(ops.attn_decode(q, kc, vc, kv, 3072, Hq, Hkv, d, d ** -0.5, workspace=ws, variant=variant), first)
    assert int(ws[:B * Hq].view(torch.int32).abs().max()) == 0
    # the counters live at a fixed offset: the same workspace serves a call with another max_kv_len (round 4: they sat behind the records,
    # so a shorter max_kv_len put them inside stale record floats and the merge never ran)
    short = torch.tensor([1200, 300], dtype=torch.int32, device=DEV)
    a = ops.attn_decode(q, kc, vc, short, 1280, Hq, Hkv, d, d ** -0.5, workspace=ws, variant=variant)
    b = ops.attn_decode(q, kc, vc, short, 1280, Hq, Hkv, d, d ** -0.5, variant=variant)
    assert torch.equal(a, b)
    assert torch.equal(ops.attn_decode(q, kc, vc, kv, 3072, Hq, Hkv, d, d ** -0.5, workspace=ws, variant=variant), first)


@pytest.mark.parametrize("variant", [0, 1])
def test_attn_decode_empty_cache_gives_zero_rows(ops, variant):
    B, Hq, Hkv, d = 3, 8, 2, 128
    g = torch.Generator().manual_seed(2)
    q = torch.randn(B, Hq * d, generator=g).bfloat16().to(DEV)
    kc = torch.randn(B, 64, Hkv * d, generator=g).bfloat16().to(DEV)
    vc = torch.randn(B, 64, Hkv * d, generator=g).bfloat16().to(DEV)
    out = torch.full((B, Hq * d), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.attn_decode(q, kc, vc, torch.tensor([0, 5, 0], dtype=torch.int32, device=DEV), 64, Hq, Hkv, d, d ** -0.5, out=out, variant=variant)
    assert bool(torch.isfinite(out.float()).all()) and float(out[0].float().abs().max()) == 0 and float(out[2].float().abs().max()) == 0
    assert float(out[1].float().abs().max()) > 0


# ------------------------------------------------------------------------------------------------ vision

def test_im2col_matches_conv(ops):
    N, H, p, C = 2, 56, 14, 32
    img = torch.randn(N, 3, H, H, generator=torch.Generator().manual_seed(1))
    w = rnd(C, 3, p, p, seed=2, scale=0.05)
    Kp = 608
    for x in (img, img.bfloat16()):
        cols = ops.im2col_patch(x.to(DEV), p, Kp)
        ref = torch.nn.functional.unfold(x.bfloat16().float(), p, stride=p).transpose(1, 2).reshape(-1, 3 * p * p)
        assert torch.equal(cols[:, :3 * p * p].cpu().float(), ref)
        assert float(cols[:, 3 * p * p:].abs().max()) == 0
    wp = torch.zeros(C, Kp, dtype=torch.bfloat16)
    wp[:, :3 * p * p] = w.reshape(C, -1)
    y = ops.gemm(cols, wp.to(DEV))
    conv = torch.nn.functional.conv2d(img.bfloat16().float(), w.float(), stride=p).flatten(2).transpose(1, 2).reshape(-1, C)
    close(y, conv, 1e-2, 2e-2, "patch embed")


@pytest.mark.parametrize("sides", [(27, 16), (27, 8), (4, 2), (4, 4)])
def test_bilinear_l2norm(ops, sides):
    si, so = sides
    f = rnd(2, si * si, 1152, seed=si)
    y = R.bilinear_reduce(f, so * so)
    close(ops.bilinear_l2norm(f.to(DEV), si, so, False), y, 8e-3, 8e-3, "bilinear")
    close(ops.bilinear_l2norm(f.to(DEV), si, so, True), R.l2_normalize(y), 1e-2, 1e-3, "bilinear+l2")


@pytest.mark.parametrize("sides", [(27, 16), (4, 2), (6, 6)])
@pytest.mark.parametrize("normalize", [True, False])
def test_bilinear_l2norm_bwd(ops, sides, normalize):
    """Backward of the 729 -> T token reduction + L2 norm (trainable tower) against autograd through the oracle ops."""
    si, so = sides
    N, C = 2, 1152 if si == 27 else 64
    x, dy = rnd(N, si * si, C, seed=1), rnd(N, so * so, C, seed=2)
    xf = x.float().requires_grad_(True)
    y = R.bilinear_reduce(xf, so * so) if si != so else xf
    if normalize:
        y = R.l2_normalize(y)
    y.backward(dy.float())
    got = ops.bilinear_l2norm_bwd(x.to(DEV), dy.to(DEV), si, so, normalize)
    close(got, xf.grad, 2e-2, 2e-2 * float(xf.grad.abs().max()), f"bilinear_l2norm bwd {sides} normalize={normalize}")


# ------------------------------------------------------------------------------------------------ optimizer

def test_adamw_and_norm(ops):
    n = 10007
    g = torch.Generator().manual_seed(1)
    p = torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    pout = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    coef = torch.tensor([0.5], device=DEV)
    for step in range(1, 4):
        gr = rnd(n, seed=10 + step)
        R.adamw_step(p, gr, m, v, step, 1e-2, 0.9, 0.95, 1e-8, 0.1, grad_scale=0.5)
        ops.adamw_shard_(pd, md, vd, gr.to(DEV), pout, 1e-2, 0.9, 0.95, 1e-8, 0.1, step, coef)
    close(pd, p, 1e-5, 1e-6, "adamw master")
    assert torch.equal(pout.cpu(), p.bfloat16()) or (pout.float().cpu() - p).abs().max() < 1e-2
    x = rnd(4099, seed=2)
    s = torch.zeros(1, device=DEV)
    ops.sumsq_(x.to(DEV), s)
    close(s[0], (x.float() ** 2).sum(), 1e-4, 1e-2, "sumsq")
    c = torch.zeros(1, device=DEV)
    ops.clip_coef(s, 1.0, 0.125, c)
    close(c[0], torch.tensor(min(1.0, 1.0 / (float((x.float() ** 2).sum()) ** 0.5 + 1e-6)) * 0.125), 1e-4, 1e-7, "clip coef")
