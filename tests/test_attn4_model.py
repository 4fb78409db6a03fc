"""CPU checks of the test infrastructure behind tests/test_attn_hostile_gpu.py (tests/attn4_model.py): the model of the d == 128 forward
stream's arithmetic agrees with the fp32 oracle, the adversarial inputs provably drive its deferred-rescale branch (so the GPU test's
"tally == model" assertion is about a branch that fires), and the arithmetic of rounds 1-4 (softmax scale folded into a re-rounded bf16
copy of q) is measurably worse than the round-5 arithmetic at large |s| -- the reason the kernels changed."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import attn4_model as M  # noqa: E402
from oracle import ref_ops as R  # noqa: E402

SCALE = 128 ** -0.5


def _oracle(q, k, v, seqlens):
    L = q.shape[1]
    valid = None if seqlens is None else (torch.arange(L)[None] < torch.tensor(seqlens)[:, None])
    return R.attention(q.transpose(1, 2).float(), k.transpose(1, 2).float(), v.transpose(1, 2).float(), valid, causal=True).transpose(1, 2)


@pytest.mark.parametrize("kind,B,L,seqlens", [("benign", 2, 300, [300, 77]), ("rising", 2, 513, [513, 400]), ("one_row", 1, 600, None),
                                              ("sink", 1, 600, None), ("cliff", 1, 640, None), ("wide", 1, 700, None), ("threshold", 3, 513, None)])
def test_model_matches_oracle_and_branch_fires(kind, B, L, seqlens):
    q, k, v = M.hostile_inputs(kind, B, L, 2, 1, seed=1)
    o, lse, cnt = M.attn4_forward_model(q, k, v, seqlens, True, SCALE)
    ref = _oracle(q, k, v, seqlens)
    ofl = M.flash_bf16_forward(q, k, v, seqlens, True, SCALE)
    for b in range(B):
        n = L if seqlens is None else seqlens[b]
        e = float((o[b, :n].float() - ref[b, :n]).abs().max())
        ef = float((ofl[b, :n].float() - ref[b, :n]).abs().max())
        assert e <= 1.5 * ef + 2e-3 * float(ref.abs().max()), (kind, b, e, ef)
        assert float(o[b, n:].float().abs().max()) == 0 if n < L else True
    if kind == "benign":
        assert int(cnt.sum()) == 0
    else:
        assert int(cnt.sum()) > 0
    if kind == "rising":
        for b in range(B):
            tw = M.last_tile_of_wave(L, L if seqlens is None else seqlens[b], True).clamp_min(0)
            assert torch.equal(cnt[b], tw[None].expand(2, -1, -1))
    if kind == "threshold":
        assert int(cnt[0].sum()) == 0 and int(cnt[1].sum()) == 0 and int(cnt[2].sum()) > 0


def test_rounds_1_to_4_arithmetic_is_worse_at_large_scores():
    q, k, v = M._base(1, 1024, 2, 1, 0, 3.0)                     # scores N(0, 9^2): |s| up to ~ 45
    ref = _oracle(q, k, v, None)
    new = M.attn4_forward_model(q, k, v, None, True, SCALE)[0]
    old = M.attn4_forward_model(q, k, v, None, True, SCALE, prescale="bf16")[0]
    fl = M.flash_bf16_forward(q, k, v, None, True, SCALE)
    rms = lambda x: float((x.float() - ref).pow(2).mean().sqrt())
    assert rms(new) <= 1.2 * rms(fl)
    assert rms(old) >= 2.0 * rms(new), (rms(old), rms(new), rms(fl))


@pytest.mark.parametrize("kind,B,L,Hq,Hkv,seqlens", [("benign", 2, 300, 2, 1, [300, 77]), ("wide", 1, 513, 4, 1, None), ("sink", 1, 400, 2, 2, None),
                                                      ("cliff", 2, 320, 2, 1, [320, 0])])
def test_backward_model_against_fp32_autograd_and_the_flash_yardstick(kind, B, L, Hq, Hkv, seqlens):
    """The model of the backward streams' arithmetic (attn4_backward_model: what tests/test_attn_hostile_gpu.py holds the kernels to, two
    bf16 steps) is itself a sane backward: on benign scores within the bf16 flash yardstick's distance of the fp32 autograd truth (x 1.5);
    on hostile scores it shows the arithmetic's known price (P ~ 1 rounded to bf16 after the lse subtraction: dv under a dominant sink) --
    bounded relative to the gradient's magnitude; zero rows beyond a sample's length; the GQA group sum in dk / dv."""
    q, k, v = M.hostile_inputs(kind, B, L, Hq, Hkv, seed=2)
    g = torch.Generator().manual_seed(9)
    do = (torch.randn(B, L, Hq, M.D, generator=g) * 0.5).to(torch.bfloat16)
    valid = None
    if seqlens is not None:
        valid = torch.arange(L)[None] < torch.tensor(seqlens)[:, None]
        do = do * valid[:, :, None, None]
    o, lse, _ = M.attn4_forward_model(q, k, v, seqlens, True, SCALE)
    dq, dk, dv = M.attn4_backward_model(q, k, v, o, do, lse, seqlens, True, SCALE)
    qf, kf, vf = (t.transpose(1, 2).float().requires_grad_(True) for t in (q, k, v))
    ref = R.attention(qf, kf, vf, valid, causal=True)
    ref = torch.nan_to_num(ref)                                   # (a sample of length 0: fully masked rows)
    (ref * do.transpose(1, 2).float()).sum().backward()
    _, fq, fk, fv = M.flash_bf16_backward(q, k, v, do, seqlens, True, SCALE)
    for name, m, y, t in (("dq", dq, fq, qf.grad.transpose(1, 2)), ("dk", dk, fk, kf.grad.transpose(1, 2)), ("dv", dv, fv, vf.grad.transpose(1, 2))):
        t = torch.nan_to_num(t)
        if seqlens is not None:
            for b in range(B):
                if seqlens[b] < L:
                    assert float(m[b, seqlens[b]:].float().abs().max()) == 0, (name, b)
        nrm = float(t.norm().clamp_min(1e-20))
        e_m, e_f = float((m.float() - t).norm()) / nrm, float((y - t).norm()) / nrm
        if kind == "benign":
            assert e_m <= 1.5 * e_f + 1e-3, (name, e_m, e_f)
        else:
            assert e_m <= 3.0 * e_f + 2e-2, (kind, name, e_m, e_f)
