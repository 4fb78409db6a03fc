"""Size-independent properties at BASELINE configs[1] sizes (LLaMA-3-8B layer geometry, 2048-token sequences), where the CPU
oracle is too slow to be the checker: linearity / additivity of the GEMM family, causality and convexity of attention,
gradient additivity over the batch, idempotence of the splice gathers.  Needs an MI355X:  pytest -m gpu"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, I, HQ, HKV, D, L = 4096, 14336, 32, 8, 128, 2048


@pytest.fixture(scope="module")
def ops():
    from metamorph_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().to(DEV)


def test_gemm_k_additivity_and_accumulate(ops):
    """x W^T over K = 4096 equals the two half-K products accumulated (ACCUMULATE epilogue), up to one bf16 rounding of the
    partial; the ping-pong kernel and the 128x128 kernel agree bit for bit."""
    M, N, K = 4096, 6144, H
    x, w = rnd(M, K, seed=1, scale=0.5), rnd(N, K, seed=2, scale=0.05)
    full = ops.gemm(x, w)
    assert torch.equal(full, ops.gemm(x, w, variant=1))
    part = torch.empty(M, N, device=DEV, dtype=torch.float32)
    ops.gemm(x[:, :K // 2], w[:, :K // 2], out=part)
    ops.gemm(x[:, K // 2:], w[:, K // 2:], out=part, accumulate=True)
    err = (part - full.float()).abs().max()
    assert float(err) <= 2 ** -7 * float(full.float().abs().max()), float(err)


def test_gemm_row_linearity(ops):
    """Scaling rows of x by powers of two scales rows of the product exactly (bf16 has no rounding under 2^k)."""
    M, N, K = 2048, 4096, I
    x, w = rnd(M, K, seed=3, scale=0.5), rnd(N, K, seed=4, scale=0.02)
    y = ops.gemm(x, w)
    y4 = ops.gemm((x.float() * 4).bfloat16(), w)
    assert torch.equal(y4, (y.float() * 4).bfloat16())


def _attn(ops, qkv, B, causal=True, seqlens=None):
    nq, nk = HQ * D, HKV * D
    return ops.attn_fwd(qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:], B, L, HQ, HKV, D, D ** -0.5, causal, seqlens)


def test_attention_causality_and_convexity(ops):
    B = 2
    qkv = rnd(B * L, (HQ + 2 * HKV) * D, seed=5, scale=0.7)
    o, lse = _attn(ops, qkv, B)
    # (1) the future cannot influence the past: perturb keys / values of the last 700 positions of sample 0
    q2 = qkv.clone()
    q2[L - 700:L, HQ * D:] = rnd(700, 2 * HKV * D, seed=6, scale=0.7)
    o2, lse2 = _attn(ops, q2, B)
    assert torch.equal(o[:L - 700], o2[:L - 700]) and torch.equal(o[L:], o2[L:])
    assert torch.equal(lse[:, :, :L - 700][0], lse2[:, :, :L - 700][0])
    # (2) rows of P sum to one: with V == 1 the output is exactly 1 wherever at least one key is visible
    q3 = qkv.clone()
    q3[:, (HQ + HKV) * D:] = 1.0
    o3, _ = _attn(ops, q3, B)
    assert float((o3.float() - 1.0).abs().max()) <= 2 ** -7
    # (3) padding: rows at or beyond seqlens are zero, earlier rows identical to the unpadded run
    sl = torch.tensor([L, 1500], dtype=torch.int32, device=DEV)
    o4, _ = _attn(ops, qkv, B, seqlens=sl)
    assert torch.equal(o4[:L + 1500], o[:L + 1500]) and float(o4[L + 1500:].abs().max()) == 0.0


def test_attention_backward_batch_additivity(ops):
    """dq/dk/dv of a batch of two samples == the per-sample results (no cross-sample term), bit for bit."""
    B = 2
    nq, nk = HQ * D, HKV * D
    qkv = rnd(B * L, (HQ + 2 * HKV) * D, seed=7, scale=0.7)
    do = rnd(B * L, HQ * D, seed=8, scale=0.5)
    o, lse = _attn(ops, qkv, B)

    def bwd(qkv_, o_, do_, lse_, b):
        d = torch.empty_like(qkv_)
        ops.attn_bwd(qkv_[:, :nq], qkv_[:, nq:nq + nk], qkv_[:, nq + nk:], o_, do_, lse_, b, L, HQ, HKV, D, D ** -0.5, True, None,
                     d[:, :nq], d[:, nq:nq + nk], d[:, nq + nk:])
        return d
    both = bwd(qkv, o, do, lse, 2)
    for b in range(2):
        rows = slice(b * L, (b + 1) * L)
        one = bwd(qkv[rows].contiguous(), o[rows].contiguous(), do[rows].contiguous(), lse[b:b + 1].contiguous(), 1)
        assert torch.equal(both[rows], one)


def test_decoder_layer_weight_gradient_accumulates_over_micro_batches():
    """One LLaMA-3-8B decoder layer: backward of two 2048-token samples one after the other (gradient buffers accumulate)
    equals the backward of the batch of both, to bf16 accumulation noise."""
    from metamorph_amd import functional as F
    from metamorph_amd.factory import LLAMA3_8B, build_model
    model = build_model(dict(LLAMA3_8B, num_hidden_layers=1), dict(num_hidden_layers=1), num_image_tokens=4, device=DEV, init_on_device=True)
    layer = model.get_model().layers[0]
    ps = list(layer.parameters())
    cos, sin = model.model.rope_tables(L, torch.device(DEV))

    def run(x, dy, B):
        meta = F.LayerMeta(B, L, HQ, HKV, D, I, 1e-5, cos, sin, None)
        xx = x.clone().requires_grad_(True)
        y = F.decoder_layer(xx, layer, meta)
        y.backward(dy)
        return xx.grad
    x, dy = rnd(2 * L, H, seed=9, scale=0.5), rnd(2 * L, H, seed=10, scale=0.1)
    for p in ps:
        p.grad = None
    dx_full = run(x, dy, 2)
    g_full = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    dx_a = run(x[:L].contiguous(), dy[:L].contiguous(), 1)
    dx_b = run(x[L:].contiguous(), dy[L:].contiguous(), 1)          # second micro-batch accumulates into the same buffers
    assert torch.equal(torch.cat([dx_a, dx_b]), dx_full)            # activations gradients carry no cross-sample term
    for p, gf in zip(ps, g_full):
        num = float((p.grad.float() - gf.float()).norm())
        den = float(gf.float().norm())
        assert num <= 1e-2 * den, (tuple(p.shape), num, den)
