"""Tensor-level wrappers over the C ABI (include/mm355.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every arithmetic op below is a
hand-written gfx950 kernel in libmm355.so reached through ctypes with raw pointers.  Nothing in this
module computes with torch ops, and nothing falls back to them.
"""
from __future__ import annotations

import os

import torch

from . import lib as _lib

BF16 = torch.bfloat16

GEMM_BIAS, GEMM_GELU_ERF, GEMM_GELU_TANH, GEMM_RESIDUAL, GEMM_ACCUMULATE, GEMM_OUT_F32 = 1, 2, 4, 8, 16, 32
GELU_ERF, GELU_TANH = 0, 1


def _L():
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Mm355Unavailable("metamorph_amd ops run only on an MI355X HIP device; got a CPU tensor "
                                        "(there is no CPU fallback)")


def _p(t):
    return 0 if t is None else t.data_ptr()


def _rows2d(t):
    """(ptr, rows, cols, ld) of a 2-D view whose last dim is contiguous."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), (t.shape, t.stride())
    return t.data_ptr(), t.shape[0], t.shape[1], t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ------------------------------------------------------------------------------------------------ GEMM

def gemm(a, b, out=None, *, bias=None, residual=None, res_row_mod=0, gelu=None, accumulate=False,
         out_f32=False, variant=0, n=None):
    """out[M,N] = epilogue(a[M,K] . b[N,K]^T).  a/b/out may be row-strided 2-D views (bf16)."""
    _chk_dev(a, b, out, bias, residual)
    assert a.dtype == BF16 and b.dtype == BF16
    pa, M, K, lda = _rows2d(a)
    pb, N, Kb, ldb = _rows2d(b)
    assert K == Kb, (a.shape, b.shape)
    if n is not None:
        N = n
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else BF16)
    po, Mo, No, ldc = _rows2d(out)
    assert Mo == M and No >= N, (out.shape, M, N)
    flags = 0
    if bias is not None:
        flags |= GEMM_BIAS
        assert bias.dtype == BF16 and bias.is_contiguous() and bias.numel() >= N
    if gelu == "erf":
        flags |= GEMM_GELU_ERF
    elif gelu == "tanh":
        flags |= GEMM_GELU_TANH
    elif gelu is not None:
        raise ValueError(gelu)
    ldr = 0
    if residual is not None:
        flags |= GEMM_RESIDUAL
        assert residual.dtype == BF16
        _, _, _, ldr = _rows2d(residual)
    if accumulate:
        flags |= GEMM_ACCUMULATE
    if out.dtype == torch.float32:
        flags |= GEMM_OUT_F32
    else:
        assert out.dtype == BF16
    rc = _L().mm355_gemm_bf16(pa, lda, pb, ldb, po, ldc, M, N, K, _p(bias), _p(residual), ldr, res_row_mod,
                              flags, variant, _stream())
    _lib.check(rc, f"mm355_gemm_bf16 M={M} N={N} K={K} lda={lda} ldb={ldb} ldc={ldc} flags={flags} variant={variant}")
    return out


def gemm_splitk(a, b, residual=None, out=None):
    """out[M, N] = a[M, K] . b[N, K]^T (+ residual), K cut into slices that run side by side (mm355_gemm_splitk_bf16): the prompt-pass form --
    a few hundred rows against one weight, where the plain kernels are a latency chain over K.  Shapes it would not split go to the plain
    kernel inside the library.  Another fp32 summation order than ops.gemm: equal to accumulation accuracy, not bit for bit."""
    _chk_dev(a, b, residual, out)
    assert a.dtype == BF16 and b.dtype == BF16
    pa, M, K, lda = _rows2d(a)
    pb, N, Kb, ldb = _rows2d(b)
    assert K == Kb
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=BF16)
    po, _, _, ldc = _rows2d(out)
    pr, ldr = 0, 0
    if residual is not None:
        pr, Mr, Nr, ldr = _rows2d(residual)
        assert (Mr, Nr) == (M, N)
    n_ws = int(_L().mm355_gemm_splitk_ws_floats(M, N, K))
    ws = torch.empty((max(n_ws, 4),), device=a.device, dtype=torch.float32)
    _lib.check(_L().mm355_gemm_splitk_bf16(pa, lda, pb, ldb, po, ldc, M, N, K, pr, ldr, ws.data_ptr(), n_ws, _stream()),
               f"mm355_gemm_splitk_bf16 M={M} N={N} K={K}")
    return out


def gemm_splitk_splits(M, N, K):
    """mm355_gemm_splitk_bf16 would cut this problem into K slices (else it forwards to the plain kernel)."""
    return int(_L().mm355_gemm_splitk_ws_floats(M, N, K)) > 0


def _splitk_ws(n_floats, device):
    return torch.empty((max(int(n_floats), 4),), device=device, dtype=torch.float32)


def gemm_splitk_norm(a, b, norm_w, eps, residual=None):
    """(c, y): c[M, N] = a . b^T (+ residual) through the split-K GEMM, y = RMSNorm(c; norm_w, eps) formed by the reduce launch
    (mm355_gemm_splitk_norm_bf16: the bits of gemm_splitk -> rmsnorm_fwd, one launch less)."""
    _chk_dev(a, b, norm_w, residual)
    pa, M, K, lda = _rows2d(a)
    pb, N, Kb, ldb = _rows2d(b)
    assert K == Kb and a.dtype == BF16 and b.dtype == BF16 and norm_w.numel() == N
    c = torch.empty((M, N), device=a.device, dtype=BF16)
    y = torch.empty((M, N), device=a.device, dtype=BF16)
    pr, ldr = 0, 0
    if residual is not None:
        pr, Mr, Nr, ldr = _rows2d(residual)
        assert (Mr, Nr) == (M, N)
    n_ws = int(_L().mm355_gemm_splitk_ws_floats(M, N, K))
    ws = _splitk_ws(n_ws, a.device)
    _lib.check(_L().mm355_gemm_splitk_norm_bf16(pa, lda, pb, ldb, c.data_ptr(), M, N, K, pr, ldr, norm_w.data_ptr(), float(eps), y.data_ptr(),
                                                ws.data_ptr(), n_ws, _stream()), f"mm355_gemm_splitk_norm_bf16 M={M} N={N} K={K}")
    return c, y


def gemm_splitk_swiglu(x, wgu, I):
    """act[M, I] = SiLU(g) * u, [g | u] = x . wgu^T through the split-K GEMM, SwiGLU in the reduce launch (mm355_gemm_splitk_swiglu_bf16: the
    bits of gemm_splitk -> swiglu_fwd)."""
    _chk_dev(x, wgu)
    px, M, K, ldx = _rows2d(x)
    pw, N, Kw, ldw = _rows2d(wgu)
    assert K == Kw and N == 2 * I and x.dtype == BF16 and wgu.dtype == BF16
    act = torch.empty((M, I), device=x.device, dtype=BF16)
    n_ws = int(_L().mm355_gemm_splitk_swiglu_ws_floats(M, I, K))
    ws = _splitk_ws(n_ws, x.device)
    _lib.check(_L().mm355_gemm_splitk_swiglu_bf16(px, ldx, pw, ldw, act.data_ptr(), act.stride(0), M, I, K, ws.data_ptr(), n_ws, _stream()),
               f"mm355_gemm_splitk_swiglu_bf16 M={M} I={I} K={K}")
    return act


def gemm_splitk_rope_append(x, wqkv, Hq, Hkv, d, cos, sin, positions, k_cache, v_cache):
    """One new q|k|v row per sequence through the split-K GEMM; RoPE at positions[m] (int32, device) and the KV-cache append in the reduce
    launch (mm355_gemm_splitk_rope_append_bf16: the bits of gemm_splitk -> rope_kv_append_).  Returns the row buffer [M, (Hq+2Hkv)*d] whose
    q columns are valid."""
    _chk_dev(x, wqkv, cos, sin, positions, k_cache, v_cache)
    px, M, K, ldx = _rows2d(x)
    pw, N, Kw, ldw = _rows2d(wqkv)
    assert K == Kw and N == (Hq + 2 * Hkv) * d and positions.dtype == torch.int32
    assert k_cache.stride() == v_cache.stride() and k_cache.stride(2) == 1
    out = torch.empty((M, N), device=x.device, dtype=BF16)
    n_ws = int(_L().mm355_gemm_splitk_ws_floats(M, N, K))
    ws = _splitk_ws(n_ws, x.device)
    _lib.check(_L().mm355_gemm_splitk_rope_append_bf16(px, ldx, pw, ldw, out.data_ptr(), out.stride(0), M, Hq, Hkv, d, K, cos.data_ptr(), sin.data_ptr(),
                                                       positions.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_cache.stride(1),
                                                       k_cache.stride(0), ws.data_ptr(), n_ws, _stream()), "mm355_gemm_splitk_rope_append_bf16")
    return out


def gemm_pair_supported(a0, b0, a1, b1):
    """Both problems fit mm355_gemm_pair_bf16 (plain NT operands, whole pairs of K tiles, 31-bit operand offsets)."""
    return (a0.shape[1] == b0.shape[1] and a1.shape[1] == b1.shape[1]
            and gemm_pp_operands_ok(a0.shape[1], a0, b0, row_major=True) and gemm_pp_operands_ok(a1.shape[1], a1, b1, row_major=True))


def gemm_pair(a0, b0, out0, acc0, a1, b1, out1, acc1):
    """out0[M0,N0] (+)= a0 . b0^T and out1[M1,N1] (+)= a1 . b1^T in ONE launch (fills the partial last wave of workgroups that
    each problem leaves on its own).  Raises Mm355Error(-2) when a problem is not eligible: check gemm_pair_supported first."""
    _chk_dev(a0, b0, out0, a1, b1, out1)
    probs = []
    for a, b, out, acc in ((a0, b0, out0, acc0), (a1, b1, out1, acc1)):
        assert a.dtype == BF16 and b.dtype == BF16
        pa, M, K, lda = _rows2d(a)
        pb, N, Kb, ldb = _rows2d(b)
        assert K == Kb, (a.shape, b.shape)
        po, Mo, No, ldc = _rows2d(out)
        assert (Mo, No) == (M, N), (out.shape, M, N)
        assert out.dtype in (BF16, torch.float32)
        flags = (GEMM_ACCUMULATE if acc else 0) | (GEMM_OUT_F32 if out.dtype == torch.float32 else 0)
        probs += [pa, lda, pb, ldb, po, ldc, M, N, K, flags]
    _lib.check(_L().mm355_gemm_pair_bf16(*probs, _stream()),
               f"mm355_gemm_pair_bf16 {tuple(a0.shape)}x{tuple(b0.shape)} + {tuple(a1.shape)}x{tuple(b1.shape)}")
    return out0, out1


def gemm_tn(at, bt, out, accumulate=False):
    """out[M,N] (+)= at[K,M]^T . bt[K,N]   (weight-gradient form, operands untransposed).  Raises Mm355Error(-2) when K % 64."""
    _chk_dev(at, bt, out)
    pa, K, M, lda = _rows2d(at)
    pb, Kb, N, ldb = _rows2d(bt)
    assert K == Kb and at.dtype == BF16 and bt.dtype == BF16
    po, Mo, No, ldc = _rows2d(out)
    assert (Mo, No) == (M, N)
    flags = (GEMM_ACCUMULATE if accumulate else 0) | (GEMM_OUT_F32 if out.dtype == torch.float32 else 0)
    _lib.check(_L().mm355_gemm_tn_bf16(pa, lda, pb, ldb, po, ldc, M, N, K, flags, _stream()), f"mm355_gemm_tn_bf16 M={M} N={N} K={K}")
    return out


def gemm_tn_supported(at, bt):
    return at.shape[0] % 64 == 0 and at.shape[1] % 8 == 0 and bt.shape[1] % 8 == 0 and at.shape[1] >= 8 and bt.shape[1] >= 8


def gemm_pp_operands_ok(K, *mats, row_major=False):
    """The ping-pong 256x256 kernel wants whole pairs of 64-wide K tiles and 31-bit byte offsets: over the whole matrix for
    contraction-major operands, over one 256-row panel for row-major ones (`row_major=True`)."""
    span = (lambda m: 256 * m.stride(0) * 2) if row_major else (lambda m: m.shape[0] * m.stride(0) * 2)
    return K >= 128 and K % 128 == 0 and all(span(m) < 2 ** 31 - 2 ** 20 for m in mats)


def gemm_nn(a, bt, out=None, residual=None, accumulate=False):
    """out[M,N] = a[M,K] . bt[K,N] (+ residual)   (input-gradient form: the weight is read as it lies in memory)."""
    _chk_dev(a, bt, out, residual)
    pa, M, K, lda = _rows2d(a)
    pb, Kb, N, ldb = _rows2d(bt)
    assert K == Kb and a.dtype == BF16 and bt.dtype == BF16
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=BF16)
    po, Mo, No, ldc = _rows2d(out)
    assert (Mo, No) == (M, N)
    flags = (GEMM_ACCUMULATE if accumulate else 0) | (GEMM_OUT_F32 if out.dtype == torch.float32 else 0)
    pr, ldr = 0, 0
    if residual is not None:
        pr, Mr, Nr, ldr = _rows2d(residual)
        assert (Mr, Nr) == (M, N)
        flags |= GEMM_RESIDUAL
    _lib.check(_L().mm355_gemm_nn_bf16(pa, lda, pb, ldb, po, ldc, M, N, K, pr, ldr, flags, _stream()),
               f"mm355_gemm_nn_bf16 M={M} N={N} K={K}")
    return out


def gemm_nn_supported(a, bt):
    M, K = a.shape
    N = bt.shape[1]
    return (N % 8 == 0 and N >= 8 and gemm_pp_operands_ok(K, a, bt)
            and ((M + 255) // 256) * ((N + 255) // 256) >= 128)


def gemv_supported(x, w):
    """mm355_gemv_bf16 takes this (x [M, K], w [N, K]) pair: up to 4 rows always; 5 - 16 rows (the MFMA form; 17 - 32 on wide weights) only while the weight and
    the x rows are addressable with 32-bit byte offsets (< 3.75 GiB; gemv_mfma_addressable in csrc/decode.hip) -- beyond that the call
    returns MM355_EUNSUPPORTED and callers use the GEMM."""
    _, M, K, ldx = _rows2d(x)
    _, N, _, ldw = _rows2d(w)
    if M > 32 or (M > 16 and not gemv_rows32_units((N + 3) // 4)):
        return False
    return M <= 4 or ((N - 1) * ldw * 2 + K * 2 < 0xf0000000 and (M - 1) * ldx * 2 + K * 2 < 0xf0000000)


def gemv_rows32_units(units):
    """17 .. 32 rows ride on the MFMA GEMVs (two x row groups per workgroup) on shapes whose workgroups each own one K range -- more than
    1280 groups of four 4-row units (gemv_mfma_shape in csrc/decode.hip): gate|up (I / 2 units), lm_head (V / 4); not q|k|v, o, down."""
    return (units + 3) // 4 > 1280


def gemv(x, w, out=None, bias=None, residual=None, gelu=None):
    """out[M,N] = x[M,K] . w[N,K]^T for M <= 16 rows (decode shape): streams the weight once at HBM rate (1 - 2 rows: vector ALU; 3 - 16: MFMA)."""
    _chk_dev(x, w, out, bias, residual)
    px, M, K, ldx = _rows2d(x)
    pw, N, Kw, ldw = _rows2d(w)
    assert K == Kw and x.dtype == BF16 and w.dtype == BF16
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=BF16)
    po, Mo, No, ldy = _rows2d(out)
    assert (Mo, No) == (M, N)
    flags = GEMM_OUT_F32 if out.dtype == torch.float32 else 0
    pr, ldr = 0, 0
    if bias is not None:
        flags |= GEMM_BIAS
    if gelu is not None:
        flags |= {"erf": GEMM_GELU_ERF, "tanh": GEMM_GELU_TANH}[gelu]
    if residual is not None:
        pr, Mr, Nr, ldr = _rows2d(residual)
        assert (Mr, Nr) == (M, N)
        flags |= GEMM_RESIDUAL
    _lib.check(_L().mm355_gemv_bf16(px, ldx, pw, ldw, po, ldy, M, N, K, _p(bias), pr, ldr, flags, _stream()), f"mm355_gemv_bf16 M={M} N={N} K={K}")
    return out


def gemv_swiglu(x, wgu, I, norm_w=None, eps=0.0, out=None):
    """act[M, I] = SiLU(g) * u, [g | u] = n . wgu^T, n = RMSNorm(x; norm_w, eps) (norm_w None: n = x); M <= 16 (decode shape)."""
    _chk_dev(x, wgu, norm_w, out)
    px, M, K, ldx = _rows2d(x)
    pw, N, Kw, ldw = _rows2d(wgu)
    assert K == Kw and N == 2 * I and x.dtype == BF16 and wgu.dtype == BF16
    out = torch.empty((M, I), device=x.device, dtype=BF16) if out is None else out
    _lib.check(_L().mm355_gemv_swiglu_bf16(px, ldx, pw, ldw, out.data_ptr(), out.stride(0), M, I, K, _p(norm_w), float(eps), _stream()),
               "mm355_gemv_swiglu_bf16")
    return out


def gemv_rope_append(x, wqkv, Hq, Hkv, d, cos, sin, positions, k_cache, v_cache, norm_w=None, eps=0.0, out=None):
    """The fused q|k|v projection of M <= 16 new rows (optionally of RMSNorm(x)), RoPE at positions[m] (int32, device), rotated k and v
    straight into cache row positions[m]; returns the row buffer [M, (Hq+2Hkv)*d] whose q columns are valid."""
    _chk_dev(x, wqkv, norm_w, cos, sin, positions, k_cache, v_cache, out)
    px, M, K, ldx = _rows2d(x)
    pw, N, Kw, ldw = _rows2d(wqkv)
    assert K == Kw and N == (Hq + 2 * Hkv) * d and positions.dtype == torch.int32
    assert k_cache.stride() == v_cache.stride() and k_cache.stride(2) == 1
    out = torch.empty((M, N), device=x.device, dtype=BF16) if out is None else out
    _lib.check(_L().mm355_gemv_rope_append_bf16(px, ldx, pw, ldw, out.data_ptr(), out.stride(0), M, Hq, Hkv, d, K, _p(norm_w), float(eps),
                                                cos.data_ptr(), sin.data_ptr(), positions.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                                k_cache.stride(1), k_cache.stride(0), _stream()), "mm355_gemv_rope_append_bf16")
    return out


def rope_kv_append_(qkv, Hq, Hkv, d, cos, sin, positions, k_cache, v_cache):
    """qkv [B, (Hq+2Hkv)*d] new rows: rotate q (in place) and k at positions[b] (int32, device); k, v -> cache row positions[b]."""
    _chk_dev(qkv, cos, sin, positions, k_cache, v_cache)
    assert positions.dtype == torch.int32 and k_cache.stride() == v_cache.stride() and k_cache.stride(2) == 1
    B = qkv.shape[0]
    _lib.check(_L().mm355_rope_kv_append(qkv.data_ptr(), qkv.stride(0), B, Hq, Hkv, d, cos.data_ptr(), sin.data_ptr(), positions.data_ptr(),
                                         k_cache.data_ptr(), v_cache.data_ptr(), k_cache.stride(1), k_cache.stride(0), _stream()),
               "mm355_rope_kv_append")
    return qkv


def attn_decode(q, k_cache, v_cache, kv_lens, max_kv_len, Hq, Hkv, d, scale, out=None, workspace=None, variant=0):
    """q [B, Hq*d]; caches [B, Lmax, Hkv*d]; kv_lens int32 [B] on the device (valid rows incl. the current one)."""
    _chk_dev(q, k_cache, v_cache, kv_lens)
    B = q.shape[0]
    assert k_cache.dim() == 3 and k_cache.shape == v_cache.shape and k_cache.stride(2) == 1 and v_cache.stride() == k_cache.stride()
    assert kv_lens.dtype == torch.int32 and max_kv_len <= k_cache.shape[1]
    out = torch.empty((B, Hq * d), device=q.device, dtype=BF16) if out is None else out
    ws = workspace
    if ws is None:
        ws = torch.zeros(int(_L().mm355_attn_decode_ws_floats(B, Hq, d, max_kv_len)), device=q.device, dtype=torch.float32)   # arrival counters start at 0
    _lib.check(_L().mm355_attn_decode_variant(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), k_cache.stride(1), k_cache.stride(0),
                                              kv_lens.data_ptr(), max_kv_len, out.data_ptr(), out.stride(0), B, Hq, Hkv, d, scale, ws.data_ptr(),
                                              int(variant), _stream()), "mm355_attn_decode")
    return out


def transpose(x, out=None, ld_out=None):
    """out[c, r] = x[r, c]"""
    _chk_dev(x, out)
    px, R, C, ld = _rows2d(x)
    if out is None:
        ldo = ld_out or ((R + 7) // 8 * 8)
        buf = torch.empty((C, ldo), device=x.device, dtype=BF16)
        out = buf[:, :R]
    po, Co, Ro, ldo = _rows2d(out)
    assert Co == C and Ro == R
    _lib.check(_L().mm355_transpose_bf16(px, ld, R, C, po, ldo, _stream()), "mm355_transpose_bf16")
    return out


def colsum_f32(dy, out_f32):
    _chk_dev(dy, out_f32)
    p, M, N, ld = _rows2d(dy)
    assert out_f32.dtype == torch.float32 and out_f32.numel() >= N
    _lib.check(_L().mm355_colsum_bf16(p, ld, M, N, out_f32.data_ptr(), _stream()), "mm355_colsum_bf16")
    return out_f32


# ------------------------------------------------------------------------------------------------ norms

def rmsnorm_fwd(x, w, eps, out=None, want_rstd=False):
    """y = w * bf16(x * rstd); want_rstd: returns (y, rstd fp32 [M]) -- what `rmsnorm_apply_t` needs in the backward pass"""
    _chk_dev(x, w)
    assert x.is_contiguous() and x.dtype == BF16 and w.dtype == BF16
    h = x.shape[-1]
    M = x.numel() // h
    out = torch.empty_like(x) if out is None else out
    if not want_rstd:
        _lib.check(_L().mm355_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, h, eps, _stream()), "mm355_rmsnorm_fwd")
        return out
    rstd = torch.empty((M,), device=x.device, dtype=torch.float32)
    _lib.check(_L().mm355_rmsnorm_fwd_rstd(x.data_ptr(), w.data_ptr(), out.data_ptr(), rstd.data_ptr(), M, h, eps, _stream()),
               "mm355_rmsnorm_fwd_rstd")
    return out, rstd


def rmsnorm_apply_t(x, w, rstd, Rp=None):
    """(w * bf16(x * rstd))^T as [h, Rp] (Rp >= M, padding columns zeroed): the normalised activations, contraction-major"""
    _chk_dev(x, w, rstd)
    assert x.is_contiguous() and x.dtype == BF16 and w.dtype == BF16 and rstd.dtype == torch.float32 and x.dim() == 2
    M, h = x.shape
    assert rstd.numel() == M
    Rp = (M + 7) // 8 * 8 if Rp is None else Rp
    assert Rp >= M and Rp % 8 == 0
    out = torch.empty((h, Rp), device=x.device, dtype=BF16)
    if Rp != M:
        out[:, M:].zero_()
    _lib.check(_L().mm355_rmsnorm_apply_t(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), M, h, out.data_ptr(), Rp, _stream()),
               "mm355_rmsnorm_apply_t")
    return out


def rmsnorm_bwd(dy, x, w, eps, dres=None, dw_f32=None, dx=None, atomic=False):
    _chk_dev(dy, x, w)
    assert dy.is_contiguous() and x.is_contiguous()
    h = x.shape[-1]
    M = x.numel() // h
    dx = torch.empty_like(x) if dx is None else dx
    ws = None
    if dw_f32 is not None and atomic is False:               # deterministic two-stage weight-gradient sum
        ws = torch.empty(int(_L().mm355_rmsnorm_bwd_ws_floats(M, h)), dtype=torch.float32, device=x.device)
    _lib.check(_L().mm355_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), _p(dres), dx.data_ptr(), _p(dw_f32), _p(ws),
                                      M, h, eps, _stream()), "mm355_rmsnorm_bwd")
    return dx


def rmsnorm_bwd_wgrad(dy, x, w, eps, w_grad, accumulate, dres=None, dx=None):
    """dx as rmsnorm_bwd; the weight gradient goes straight into the bf16 buffer `w_grad` ((+)= when accumulate)."""
    _chk_dev(dy, x, w, w_grad)
    assert dy.is_contiguous() and x.is_contiguous() and w_grad.is_contiguous() and w_grad.dtype == BF16
    h = x.shape[-1]
    M = x.numel() // h
    dx = torch.empty_like(x) if dx is None else dx
    ws = torch.empty(int(_L().mm355_rmsnorm_bwd_ws_floats(M, h)), dtype=torch.float32, device=x.device)
    _lib.check(_L().mm355_rmsnorm_bwd_wgrad(dy.data_ptr(), x.data_ptr(), w.data_ptr(), _p(dres), dx.data_ptr(), w_grad.data_ptr(), int(bool(accumulate)),
                                            ws.data_ptr(), M, h, eps, _stream()), "mm355_rmsnorm_bwd_wgrad")
    return dx


def layernorm_fwd(x, w, b, eps, out=None):
    _chk_dev(x, w, b)
    assert x.is_contiguous()
    h = x.shape[-1]
    M = x.numel() // h
    out = torch.empty_like(x) if out is None else out
    _lib.check(_L().mm355_layernorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, h, eps, _stream()),
               "mm355_layernorm_fwd")
    return out


def layernorm_bwd(dy, x, w, eps, dw_f32, db_f32, dres=None, dx=None):
    """dx = (dres or 0) + d layernorm / dx; dw_f32 += sum dy*xhat, db_f32 += sum dy (deterministic two-stage sums)."""
    _chk_dev(dy, x, w, dw_f32, db_f32)
    assert dy.is_contiguous() and x.is_contiguous() and dw_f32.dtype == torch.float32 and db_f32.dtype == torch.float32
    h = x.shape[-1]
    M = x.numel() // h
    dx = torch.empty_like(x) if dx is None else dx
    ws = torch.empty(int(_L().mm355_layernorm_bwd_ws_floats(M, h)), dtype=torch.float32, device=x.device)
    _lib.check(_L().mm355_layernorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), _p(dres), dx.data_ptr(), dw_f32.data_ptr(), db_f32.data_ptr(),
                                        ws.data_ptr(), M, h, eps, _stream()), "mm355_layernorm_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ rope / attention

def rope_table(L, d, theta, device):
    cos = torch.empty((L, d), device=device, dtype=BF16)
    sin = torch.empty((L, d), device=device, dtype=BF16)
    _chk_dev(cos)
    _lib.check(_L().mm355_rope_table(cos.data_ptr(), sin.data_ptr(), L, d, theta, _stream()), "mm355_rope_table")
    return cos, sin


def rope_table_freq(L, d, inv_freq, attention_scaling, device):
    """cos / sin [L, d] bf16 from host-computed inverse frequencies (float32 [d/2]: numpy array or tensor) -- metamorph_amd.rope."""
    inv = torch.as_tensor(inv_freq, dtype=torch.float32).reshape(-1).to(device)
    assert inv.numel() * 2 == d, (inv.numel(), d)
    cos = torch.empty((L, d), device=device, dtype=BF16)
    sin = torch.empty((L, d), device=device, dtype=BF16)
    _chk_dev(cos, inv)
    _lib.check(_L().mm355_rope_table_freq(cos.data_ptr(), sin.data_ptr(), L, d, inv.data_ptr(), float(attention_scaling), _stream()),
               "mm355_rope_table_freq")
    return cos, sin


def rope_qk_(qkv, B, L, Hq, Hkv, d, cos, sin, inverse=False, pos_offset=None):
    """In place on the q (first Hq*d columns) and k (next Hkv*d columns) blocks of qkv [B*L, ld]; pos_offset (int32 [B], device):
    position of row (b, l) is l + pos_offset[b] (the caller guarantees the tables are long enough)."""
    _chk_dev(qkv, cos, sin, pos_offset)
    p, M, _, ld = _rows2d(qkv)
    assert M == B * L and cos.shape[0] >= L
    if pos_offset is None:
        _lib.check(_L().mm355_rope_qk(p, ld, B, L, Hq, Hkv, d, cos.data_ptr(), sin.data_ptr(), int(inverse), _stream()), "mm355_rope_qk")
    else:
        assert pos_offset.dtype == torch.int32 and pos_offset.numel() == B
        _lib.check(_L().mm355_rope_qk_pos(p, ld, B, L, Hq, Hkv, d, cos.data_ptr(), sin.data_ptr(), pos_offset.data_ptr(), int(inverse), _stream()),
                   "mm355_rope_qk_pos")
    return qkv


def gemm_rope_supported(x, wqkv, Hq, Hkv, d, cos):
    """mm355_gemm_rope_bf16 takes this problem (head size 128, whole 256-column tiles, at least one wave of 256 x 256 tiles)."""
    M, K = x.shape
    N = wqkv.shape[0]
    return (d == 128 and N == (Hq + 2 * Hkv) * d and N % 256 == 0 and K >= 128 and K % 128 == 0 and wqkv.shape[1] == K
            and wqkv.is_contiguous() and x.is_contiguous() and cos.shape[1] == 128 and ((M + 255) // 256) * (N // 256) >= 200)


def gemm_rope(x, wqkv, B, L, Hq, Hkv, d, cos, sin, pos_offset=None):
    """qkv [B*L, (Hq + 2 Hkv) d] = fused q|k|v projection + RoPE on the q / k blocks: the bits of gemm(x, wqkv) followed by rope_qk_."""
    _chk_dev(x, wqkv, cos, sin, pos_offset)
    M, K = x.shape
    N = wqkv.shape[0]
    assert M == B * L and x.dtype == BF16 and wqkv.dtype == BF16 and cos.shape[0] >= L
    if pos_offset is not None:
        assert pos_offset.dtype == torch.int32 and pos_offset.numel() == B
    qkv = torch.empty((M, N), device=x.device, dtype=BF16)
    _lib.check(_L().mm355_gemm_rope_bf16(x.data_ptr(), K, wqkv.data_ptr(), K, qkv.data_ptr(), N, cos.data_ptr(), sin.data_ptr(), _p(pos_offset),
                                         M, N, K, L, (Hq + Hkv) * d, _stream()), f"mm355_gemm_rope_bf16 M={M} N={N} K={K}")
    return qkv


def attn_fwd(q2d, k2d, v2d, B, L, Hq, Hkv, d, scale, causal, seqlens=None, out=None, variant=0):
    """q2d/k2d/v2d: [B*L, ld] views starting at the q / k / v column blocks (k and v share the leading dimension).
    variant != 0 (tests / tools only) picks the kernel generation: mm355_attn_fwd_variant in include/mm355.h."""
    _chk_dev(q2d, k2d, v2d)
    pq, M, _, ldq = _rows2d(q2d)
    pk, _, _, ldk = _rows2d(k2d)
    pv, _, _, ldv = _rows2d(v2d)
    assert ldv == ldk and M == B * L
    if out is None:
        out = torch.empty((M, Hq * d), device=q2d.device, dtype=BF16)
    lse = torch.empty((B, Hq, L), device=q2d.device, dtype=torch.float32)
    if variant:
        _lib.check(_L().mm355_attn_fwd_variant(pq, pk, pv, ldq, ldk, out.data_ptr(), out.stride(0), lse.data_ptr(), _p(seqlens),
                                               B, L, Hq, Hkv, d, scale, int(causal), int(variant), _stream()), "mm355_attn_fwd_variant")
        return out, lse
    _lib.check(_L().mm355_attn_fwd(pq, pk, pv, ldq, ldk, out.data_ptr(), out.stride(0), lse.data_ptr(), _p(seqlens),
                                   B, L, Hq, Hkv, d, scale, int(causal), _stream()), "mm355_attn_fwd")
    return out, lse


def attn_fwd_debug(q2d, k2d, v2d, B, L, Hq, Hkv, d, scale, causal, seqlens=None, variant=4):
    """attn_fwd through mm355_attn_fwd_debug (d == 128 stream, variant 4 or its serialised twin 41) -> (o, lse, rescale_counts):
    rescale_counts int32 [B, Hq, ceil(L / 256), 4] = how often each wave took the deferred-rescale branch.  Tests only."""
    _chk_dev(q2d, k2d, v2d)
    pq, M, _, ldq = _rows2d(q2d)
    pk, _, _, ldk = _rows2d(k2d)
    pv, _, _, ldv = _rows2d(v2d)
    assert ldv == ldk and M == B * L
    out = torch.empty((M, Hq * d), device=q2d.device, dtype=BF16)
    lse = torch.empty((B, Hq, L), device=q2d.device, dtype=torch.float32)
    counts = torch.zeros((B, Hq, (L + 255) // 256, 4), device=q2d.device, dtype=torch.int32)
    _lib.check(_L().mm355_attn_fwd_debug(pq, pk, pv, ldq, ldk, out.data_ptr(), out.stride(0), lse.data_ptr(), _p(seqlens),
                                         B, L, Hq, Hkv, d, scale, int(causal), int(variant), counts.data_ptr(), _stream()), "mm355_attn_fwd_debug")
    return out, lse, counts


def attn_bwd_rope_supported(d):
    """The d == 128 kernels can rotate dq / dk back through RoPE in their epilogues (mm355_attn_bwd_rope)."""
    return d == 128


def attn_bwd(q2d, k2d, v2d, o, d_o, lse, B, L, Hq, Hkv, d, scale, causal, seqlens, dq2d, dk2d, dv2d, rope=None, variant=0):
    """Writes dq / dk / dv (bf16) into the given column-block views.  rope = (cos, sin, pos_offset | None): q / k are post-RoPE and dq /
    dk are wanted as gradients of the PRE-RoPE projections -- the inverse rotation runs in the kernels' epilogues (d == 128 only).
    variant != 0 (tests / tools only) picks the kernel generation: mm355_attn_bwd_variant in include/mm355.h."""
    _chk_dev(q2d, k2d, v2d, o, d_o, dq2d, dk2d, dv2d)
    pq, M, _, ldq = _rows2d(q2d)
    pk, _, _, ldk = _rows2d(k2d)
    pv, _, _, ldv = _rows2d(v2d)
    assert ldv == ldk
    assert o.is_contiguous() and d_o.is_contiguous()
    delta = torch.empty((B, Hq, L), device=o.device, dtype=torch.float32)
    _lib.check(_L().mm355_attn_bwd_prep(o.data_ptr(), d_o.data_ptr(), o.stride(0), delta.data_ptr(), B, L, Hq, d, _stream()),
               "mm355_attn_bwd_prep")
    n_ws = int(_L().mm355_attn_bwd_ws_floats(B, L, Hq, Hkv, d, max(ldq, ldk, d_o.stride(0))))
    if variant == 2 and Hq != Hkv:                           # the generic kernels sum a GQA group from fp32 per-query-head partials
        n_ws = max(n_ws, 2 * B * L * Hq * d)
    ws = torch.empty(n_ws, device=o.device, dtype=torch.float32) if n_ws else None     # d = 128: per-row constants; generic-d GQA: partials
    pdq, _, _, lddq = _rows2d(dq2d)
    pdk, _, _, lddk = _rows2d(dk2d)
    pdv, _, _, lddv = _rows2d(dv2d)
    assert lddk == lddv
    cos = sin = pos_offset = None
    if rope is not None:
        cos, sin, pos_offset = rope
        _chk_dev(cos, sin, pos_offset)
        assert cos.shape[1] == d and cos.shape[0] >= L and (pos_offset is None or (pos_offset.dtype == torch.int32 and pos_offset.numel() == B))
    _lib.check(_L().mm355_attn_bwd_variant(pq, pk, pv, ldq, ldk, d_o.data_ptr(), d_o.stride(0), lse.data_ptr(), delta.data_ptr(), _p(seqlens),
                                           pdq, lddq, pdk, pdv, lddk, B, L, Hq, Hkv, d, scale, int(causal), _p(cos), _p(sin), _p(pos_offset),
                                           _p(ws), int(variant), _stream()), "mm355_attn_bwd_variant")


def cast_f32_to_bf16_2d(src_f32, dst2d):
    _chk_dev(src_f32, dst2d)
    pd, R, C, ldd = _rows2d(dst2d)
    assert src_f32.dim() == 2 and src_f32.shape == (R, C) and src_f32.stride(1) == 1
    _lib.check(_L().mm355_cast_f32_bf16_2d(src_f32.data_ptr(), src_f32.stride(0), pd, ldd, R, C, _stream()), "mm355_cast_f32_bf16_2d")
    return dst2d


# ------------------------------------------------------------------------------------------------ elementwise

def gemm_swiglu_supported(x, wgu, I):
    """mm355_gemm_swiglu_bf16 takes this problem (whole 128-channel tiles and K-tile pairs, a launch of at least one wave of
    256 x 256 tiles -- smaller problems go through mm355_gemm_bf16's small-tile kernels + mm355_swiglu_fwd)."""
    M, K = x.shape
    tiles = ((M + 255) // 256) * ((2 * I + 255) // 256)
    return (I % 128 == 0 and K >= 128 and K % 128 == 0 and wgu.shape[0] == 2 * I and wgu.is_contiguous() and x.is_contiguous()
            and tiles >= 200 and (2 * I * K + K) * 2 < 0x7fffffff)


def gemm_swiglu(x, wgu, I):
    """(gu [M, 2I], act [M, I]) = fused gate|up GEMM + SwiGLU: the bits of gemm(x, wgu) and swiglu_fwd(gu, I) in one launch."""
    _chk_dev(x, wgu)
    M, K = x.shape
    assert x.dtype == BF16 and wgu.dtype == BF16 and wgu.shape == (2 * I, K)
    gu = torch.empty((M, 2 * I), device=x.device, dtype=BF16)
    act = torch.empty((M, I), device=x.device, dtype=BF16)
    _lib.check(_L().mm355_gemm_swiglu_bf16(x.data_ptr(), K, wgu.data_ptr(), K, gu.data_ptr(), 2 * I, act.data_ptr(), I, M, I, K, _stream()),
               f"mm355_gemm_swiglu_bf16 M={M} I={I} K={K}")
    return gu, act


def gemm_swiglu_bwd_supported(dy, wdT, gu, I):
    """mm355_gemm_swiglu_bwd_bf16 takes this problem (and it is big enough for the 256 x 256 ping-pong tiles)."""
    M, K = dy.shape
    tiles = ((M + 255) // 256) * ((I + 255) // 256)
    return (I % 64 == 0 and M % 64 == 0 and K >= 128 and K % 128 == 0 and wdT.shape[0] == I and wdT.shape[1] == K and wdT.is_contiguous()
            and dy.is_contiguous() and gu.is_contiguous() and gu.shape == (M, 2 * I) and tiles >= 200)


def gemm_swiglu_bwd(dy, wdT, gu, I):
    """(dgu [M, 2I], actT [I, M], dguT [2I, M]) = SwiGLU backward of d act = dy . wdT^T, fused into that GEMM's epilogue: the bits of
    gemm(dy, wdT) followed by swiglu_bwd_t(gu, d act, I), without the d act round trip through HBM."""
    _chk_dev(dy, wdT, gu)
    M, K = dy.shape
    assert dy.dtype == BF16 and wdT.dtype == BF16 and gu.dtype == BF16
    dgu = torch.empty_like(gu)
    actT = torch.empty((I, M), device=gu.device, dtype=BF16)
    dguT = torch.empty((2 * I, M), device=gu.device, dtype=BF16)
    _lib.check(_L().mm355_gemm_swiglu_bwd_bf16(dy.data_ptr(), K, wdT.data_ptr(), K, gu.data_ptr(), 2 * I, dgu.data_ptr(), 2 * I,
                                               actT.data_ptr(), dguT.data_ptr(), M, M, I, K, _stream()),
               f"mm355_gemm_swiglu_bwd_bf16 M={M} I={I} K={K}")
    return dgu, actT, dguT


def swiglu_fwd(gu, I):
    _chk_dev(gu)
    assert gu.is_contiguous() and gu.shape[-1] == 2 * I
    M = gu.numel() // (2 * I)
    act = torch.empty((M, I), device=gu.device, dtype=BF16)
    _lib.check(_L().mm355_swiglu_fwd(gu.data_ptr(), act.data_ptr(), M, I, _stream()), "mm355_swiglu_fwd")
    return act


def swiglu_bwd(gu, dact, I, want_act=True):
    _chk_dev(gu, dact)
    assert gu.is_contiguous() and dact.is_contiguous()
    M = gu.numel() // (2 * I)
    dgu = torch.empty_like(gu)
    act = torch.empty((M, I), device=gu.device, dtype=BF16) if want_act else None
    _lib.check(_L().mm355_swiglu_bwd(gu.data_ptr(), dact.data_ptr(), dgu.data_ptr(), _p(act), M, I, _stream()), "mm355_swiglu_bwd")
    return dgu, act


def swiglu_bwd_t(gu, dact, I):
    """(dgu [M,2I], actT [I,M], dguT [2I,M]) -- SwiGLU backward that also writes the transposed copies for the dW GEMMs."""
    _chk_dev(gu, dact)
    assert gu.is_contiguous() and dact.is_contiguous()
    M = gu.numel() // (2 * I)
    dgu = torch.empty_like(gu)
    actT = torch.empty((I, M), device=gu.device, dtype=BF16)
    dguT = torch.empty((2 * I, M), device=gu.device, dtype=BF16)
    _lib.check(_L().mm355_swiglu_bwd_t(gu.data_ptr(), dact.data_ptr(), dgu.data_ptr(), actT.data_ptr(), dguT.data_ptr(), M, I, _stream()),
               "mm355_swiglu_bwd_t")
    return dgu, actT, dguT


def gelu_fwd(x, kind=GELU_ERF):
    _chk_dev(x)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.check(_L().mm355_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), kind, _stream()), "mm355_gelu_fwd")
    return y


def gelu_bwd(x, dy, kind=GELU_ERF):
    _chk_dev(x, dy)
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    _lib.check(_L().mm355_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), kind, _stream()), "mm355_gelu_bwd")
    return dx


def scale_(x, s_dev=None, s_host=1.0):
    _chk_dev(x)
    assert x.is_contiguous() and x.dtype == BF16
    if s_dev is not None:
        assert s_dev.dtype == torch.float32
    _lib.check(_L().mm355_scale_bf16(x.data_ptr(), x.numel(), _p(s_dev), s_host, _stream()), "mm355_scale_bf16")
    return x


def axpy_(y, x, s_dev=None, s_host=1.0, accumulate=True):
    """y (+)= s * x ; y bf16, x bf16 or f32 (contiguous, same numel)."""
    _chk_dev(y, x)
    assert y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel() and y.dtype == BF16
    if x.dtype == torch.float32:
        if s_dev is None:
            _lib.check(_L().mm355_axpy_f32_to_bf16(y.data_ptr(), x.data_ptr(), y.numel(), s_host, int(accumulate), _stream()), "mm355_axpy_f32_to_bf16")
        else:
            assert s_dev.dtype == torch.float32
            _lib.check(_L().mm355_axpy_f32_to_bf16_dev(y.data_ptr(), x.data_ptr(), y.numel(), s_dev.data_ptr(), s_host, int(accumulate), _stream()),
                       "mm355_axpy_f32_to_bf16_dev")
    else:
        _lib.check(_L().mm355_axpy_bf16(y.data_ptr(), x.data_ptr(), y.numel(), _p(s_dev), s_host, int(accumulate), _stream()), "mm355_axpy_bf16")
    return y


# ------------------------------------------------------------------------------------------------ losses

def linear_ce(hidden, rows, targets, weight, need_dh=True, need_dw=True, dw_f32=None):
    """mm355_linear_ce: mean NLL of softmax(hidden[rows] . weight^T) at `targets`, plus d loss / d hidden[rows] (compact [n, h] bf16) and
    d loss / d weight ([V, h]; fp32 when more than one 8192-row chunk accumulates into it, else bf16) -> (loss f32 [1], d_hidden | None,
    dW | None).  hidden [M, h] bf16 (row-strided view allowed); rows / targets int32 [n] on the device (rows None: the first n rows)."""
    _chk_dev(hidden, rows, targets, weight)
    ph, M, h, ldh = _rows2d(hidden)
    pw, V, hw, ldw = _rows2d(weight)
    n = int(targets.numel())
    assert hw == h and hidden.dtype == BF16 and weight.dtype == BF16 and targets.dtype == torch.int32 and n > 0
    assert rows is None or (rows.dtype == torch.int32 and rows.numel() == n)
    dev = hidden.device
    if dw_f32 is None:
        dw_f32 = n > 8192
    loss = torch.empty((1,), device=dev, dtype=torch.float32)
    dh = torch.empty((n, h), device=dev, dtype=BF16) if need_dh else None
    dw = torch.empty((V, h), device=dev, dtype=torch.float32 if dw_f32 else BF16) if need_dw else None
    nbytes = int(_L().mm355_linear_ce_ws_bytes(n, V, h, int(rows is not None), int(need_dh), int(need_dw)))
    ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    _lib.check(_L().mm355_linear_ce(ph, ldh, _p(rows), targets.data_ptr(), n, pw, ldw, V, h, loss.data_ptr(), _p(dh), _p(dw), int(bool(dw_f32)),
                                    ws.data_ptr(), nbytes, _stream()), f"mm355_linear_ce n={n} V={V} h={h}")
    return loss, dh, dw


def _row_ws(R, device, deterministic=True):
    """R floats for the per-row loss values (summed in a fixed order -> bit-reproducible loss scalars); None = atomics (tests only)."""
    return torch.empty((R,), device=device, dtype=torch.float32) if deterministic else None


def ce_rows_(logits2d, targets_i32, V, grad_scale, loss_sum_f32, deterministic=True):
    _chk_dev(logits2d, targets_i32, loss_sum_f32)
    p, R, _, ld = _rows2d(logits2d)
    assert targets_i32.dtype == torch.int32 and targets_i32.numel() >= R
    ws = _row_ws(R, logits2d.device, deterministic)
    _lib.check(_L().mm355_ce_rows(p, ld, targets_i32.data_ptr(), R, V, grad_scale, loss_sum_f32.data_ptr(), _p(ws), _stream()), "mm355_ce_rows")


def cosine_loss(pred_raw, target, normalize, want_grad=True):
    _chk_dev(pred_raw, target)
    assert pred_raw.is_contiguous() and target.is_contiguous() and pred_raw.shape == target.shape
    R, C = pred_raw.shape
    cos_sum = torch.zeros((1,), device=pred_raw.device, dtype=torch.float32)
    dpred = torch.empty_like(pred_raw) if want_grad else None
    _lib.check(_L().mm355_cosine_loss(pred_raw.data_ptr(), target.data_ptr(), R, C, int(normalize), cos_sum.data_ptr(), _p(dpred),
                                      _p(_row_ws(R, pred_raw.device)), _stream()), "mm355_cosine_loss")
    return cos_sum, dpred


def mean_abs_loss(pred, target, want_grad=True):
    """(sum |target - pred| as a 1-element f32 tensor, d mean|t - p| / d pred or None)  -- reference `mse_loss_fn`."""
    _chk_dev(pred, target)
    assert pred.is_contiguous() and target.is_contiguous() and pred.shape == target.shape and pred.dtype == BF16 and target.dtype == BF16
    R, C = pred.shape
    abs_sum = torch.zeros((1,), device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred) if want_grad else None
    _lib.check(_L().mm355_mean_abs_loss(pred.data_ptr(), target.data_ptr(), R, C, abs_sum.data_ptr(), _p(dpred), _p(_row_ws(R, pred.device)),
                                        _stream()), "mm355_mean_abs_loss")
    return abs_sum, dpred


def soft_ce_loss(pred_raw, target, normalize, temperature=0.07, want_grad=True):
    """(sum_r -sum_j t log(softmax(u / temperature) + 1e-10), d (that / R) / d pred_raw or None)."""
    _chk_dev(pred_raw, target)
    assert pred_raw.is_contiguous() and target.is_contiguous() and pred_raw.shape == target.shape
    assert pred_raw.dtype == BF16 and target.dtype == BF16
    R, C = pred_raw.shape
    loss_sum = torch.zeros((1,), device=pred_raw.device, dtype=torch.float32)
    dpred = torch.empty_like(pred_raw) if want_grad else None
    _lib.check(_L().mm355_soft_ce_loss(pred_raw.data_ptr(), target.data_ptr(), R, C, int(bool(normalize)), float(temperature),
                                       loss_sum.data_ptr(), _p(dpred), _p(_row_ws(R, pred_raw.device)), _stream()), "mm355_soft_ce_loss")
    return loss_sum, dpred


def softmax_rows(x2d, temperature=0.07):
    _chk_dev(x2d)
    assert x2d.is_contiguous() and x2d.dtype == BF16 and x2d.dim() == 2
    y = torch.empty_like(x2d)
    _lib.check(_L().mm355_softmax_rows(x2d.data_ptr(), y.data_ptr(), x2d.shape[0], x2d.shape[1], float(temperature), _stream()), "mm355_softmax_rows")
    return y


def softmax_rows_bwd(y2d, dy2d, temperature=0.07):
    _chk_dev(y2d, dy2d)
    assert y2d.is_contiguous() and dy2d.is_contiguous() and y2d.shape == dy2d.shape and y2d.dtype == BF16 and dy2d.dtype == BF16
    dx = torch.empty_like(y2d)
    _lib.check(_L().mm355_softmax_rows_bwd(y2d.data_ptr(), dy2d.data_ptr(), dx.data_ptr(), y2d.shape[0], y2d.shape[1], float(temperature),
                                           _stream()), "mm355_softmax_rows_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ splice / rows

def splice_gather(embed, proj2d, src_i32, h):
    _chk_dev(embed, proj2d, src_i32)
    rows = src_i32.numel()
    out = torch.empty((rows, h), device=embed.device, dtype=BF16)
    assert embed.is_contiguous() and (proj2d is None or proj2d.is_contiguous())
    _lib.check(_L().mm355_splice_gather(embed.data_ptr(), _p(proj2d), src_i32.data_ptr(), out.data_ptr(), rows, h, _stream()), "mm355_splice_gather")
    return out


def rows_gather(x2d, idx_i32, out=None):
    _chk_dev(x2d, idx_i32)
    p, _, h, ld = _rows2d(x2d)
    R = idx_i32.numel()
    if out is None:
        out = torch.empty((R, h), device=x2d.device, dtype=BF16)
    _lib.check(_L().mm355_rows_gather(p, ld, idx_i32.data_ptr(), out.data_ptr(), out.stride(0), R, h, _stream()), "mm355_rows_gather")
    return out


def rows_scatter_add_(dst2d, src2d, idx_i32):
    _chk_dev(dst2d, src2d, idx_i32)
    ps, R, h, lds = _rows2d(src2d)
    pd, _, _, ldd = _rows2d(dst2d)
    _lib.check(_L().mm355_rows_scatter_add(ps, lds, idx_i32.data_ptr(), pd, ldd, R, h, _stream()), "mm355_rows_scatter_add")
    return dst2d


def embed_grad_(dembed, dout2d, tok_i32, seg_i32, pos_i32, accumulate):
    _chk_dev(dembed, dout2d)
    assert dout2d.is_contiguous() and dembed.is_contiguous()
    n_seg = tok_i32.numel()
    if n_seg == 0:
        return dembed
    _lib.check(_L().mm355_embed_grad(dout2d.data_ptr(), tok_i32.data_ptr(), seg_i32.data_ptr(), pos_i32.data_ptr(), n_seg,
                                     dembed.data_ptr(), dembed.shape[1], int(accumulate), _stream()), "mm355_embed_grad")
    return dembed


# ------------------------------------------------------------------------------------------------ vision

def im2col_patch(images, p, Kp):
    _chk_dev(images)
    assert images.is_contiguous() and images.dim() == 4 and images.shape[1] == 3
    N, _, H, W = images.shape
    is_f32 = images.dtype == torch.float32
    assert is_f32 or images.dtype == BF16
    out = torch.empty((N * (H // p) * (W // p), Kp), device=images.device, dtype=BF16)
    _lib.check(_L().mm355_im2col_patch(images.data_ptr(), int(is_f32), N, H, W, p, out.data_ptr(), Kp, _stream()), "mm355_im2col_patch")
    return out


def bilinear_l2norm(feat, side_in, side_out, normalize):
    _chk_dev(feat)
    assert feat.is_contiguous()
    N, P, C = feat.shape
    assert P == side_in * side_in
    out = torch.empty((N, side_out * side_out, C), device=feat.device, dtype=BF16)
    _lib.check(_L().mm355_bilinear_l2norm(feat.data_ptr(), out.data_ptr(), N, side_in, side_out, C, int(normalize), _stream()), "mm355_bilinear_l2norm")
    return out


def bilinear_l2norm_bwd(x, dy, side_in, side_out, normalize):
    """Gradient of bilinear_l2norm w.r.t. x [N, side_in^2, C] (bf16) from dy [N, side_out^2, C]; returns bf16."""
    _chk_dev(x, dy)
    assert x.is_contiguous() and dy.is_contiguous()
    N, _, C = x.shape
    acc = torch.zeros(x.shape, device=x.device, dtype=torch.float32)
    _lib.check(_L().mm355_bilinear_l2norm_bwd(x.data_ptr(), dy.data_ptr(), acc.data_ptr(), N, side_in, side_out, C, int(bool(normalize)),
                                              _stream()), "mm355_bilinear_l2norm_bwd")
    out = torch.empty_like(x)
    cast_f32_to_bf16_2d(acc.view(-1, C), out.view(-1, C))
    return out


# ------------------------------------------------------------------------------------------------ optimizer

def adamw_shard_(p32, m, v, g, p_out, lr, beta1, beta2, eps, wd, step, grad_scale_dev=None):
    _chk_dev(p32, m, v, g, p_out)
    n = p32.numel()
    assert m.numel() == n and v.numel() == n and g.numel() == n and p_out.numel() == n
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    _lib.check(_L().mm355_adamw_shard(p32.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), p_out.data_ptr(), n, lr, beta1, beta2,
                                      eps, wd, bc1, bc2, _p(grad_scale_dev), _stream()), "mm355_adamw_shard")


SUMSQ_PARTIALS = 2048            # MM355_SUMSQ_PARTIALS (include/mm355.h)
_sumsq_partials = {}


def sumsq_(x, out_f32, partials=None):
    """out_f32[0] += sum(x^2), deterministic (two launches, no atomics).  `partials`: fp32 scratch of SUMSQ_PARTIALS elements; by
    default one buffer per (device, stream), so launches on different streams never share it."""
    _chk_dev(x, out_f32)
    assert x.is_contiguous()
    if partials is None:
        key = (x.device, _stream())
        partials = _sumsq_partials.get(key)
        if partials is None:
            partials = _sumsq_partials[key] = torch.empty(SUMSQ_PARTIALS, dtype=torch.float32, device=x.device)
    assert partials.dtype == torch.float32 and partials.numel() >= SUMSQ_PARTIALS and partials.device == x.device
    _lib.check(_L().mm355_sumsq_bf16(x.data_ptr(), x.numel(), out_f32.data_ptr(), partials.data_ptr(), _stream()), "mm355_sumsq_bf16")
    return out_f32


def clip_coef(sumsq_f32, max_norm, pre_scale, out_f32):
    _chk_dev(sumsq_f32, out_f32)
    _lib.check(_L().mm355_clip_coef(sumsq_f32.data_ptr(), max_norm, pre_scale, out_f32.data_ptr(), _stream()), "mm355_clip_coef")
    return out_f32
