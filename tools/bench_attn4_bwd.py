#!/usr/bin/env python3
"""attn4 backward streams (variant 4; 41 = serialised) against the shipped attn3 kernels (variant 3) and fp32 autograd on the device:
correctness on the d == 128 cases of tests/test_kernels_gpu.py + full-size shapes, bit-equality of the placed and the serialised streams,
RoPE-fused epilogues against variant 3, then timings.   python tools/bench_attn4_bwd.py [--quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd import ops

DEV = "cuda"


def _base_variant():
    """the comparison kernels: attn3 (round 2's d == 128 kernels) when the library was built with MM355_LEGACY_VARIANTS=1, else the generic attn2"""
    from metamorph_amd.lib import Mm355Error
    x = torch.zeros(64, 3 * 128, device=DEV, dtype=torch.bfloat16)
    try:
        ops.attn_fwd(x[:, :128], x[:, 128:256], x[:, 256:], 1, 64, 1, 1, 128, 128 ** -0.5, True, None, variant=3)
        return 3
    except Mm355Error:
        return 2


BASE = None


def ref_grads(q, k, v, do, seqlens, causal):
    """fp32 autograd on the device; q [B, L, Hq, d] ..."""
    B, L, Hq, d = q.shape
    Hkv = k.shape[2]
    rep = Hq // Hkv
    q, k, v = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = torch.zeros_like(q)
    for b in range(B):
        n = seqlens[b] if seqlens else L
        kk = k[b].repeat_interleave(rep, dim=1)                # [L, Hq, d]
        vv = v[b].repeat_interleave(rep, dim=1)
        s = torch.einsum("qhd,khd->hqk", q[b], kk) * d ** -0.5
        mask = torch.zeros(L, L, dtype=torch.bool, device=q.device)
        if causal:
            mask |= ~torch.ones(L, L, dtype=torch.bool, device=q.device).tril()
        mask[:, n:] = True
        s = s.masked_fill(mask[None], float("-inf"))
        p = torch.softmax(s, -1)
        ob = torch.einsum("hqk,khd->qhd", p, vv)
        out = out + torch.nn.functional.pad(ob[None], (0, 0, 0, 0, 0, 0, b, B - 1 - b))
    (out * do).sum().backward()
    return q.grad, k.grad, v.grad


def run_case(B, L, Hq, Hkv, causal, seqlens, seed=0, rope=False):
    d = 128
    g = torch.Generator(device="cpu").manual_seed(seed)
    ld = (Hq + 2 * Hkv) * d
    qkv = (torch.randn(B * L, ld, generator=g) * 0.7).bfloat16().to(DEV)
    do = (torch.randn(B * L, Hq * d, generator=g) * 0.5).bfloat16().to(DEV)
    nq, nk = Hq * d, Hkv * d
    if seqlens:
        valid = (torch.arange(L)[None] < torch.tensor(seqlens)[:, None]).to(DEV)
        do = (do.view(B, L, -1) * valid[:, :, None]).reshape(B * L, -1).contiguous()
    q2, k2, v2 = qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:]
    sl = torch.tensor(seqlens, dtype=torch.int32, device=DEV) if seqlens else None
    o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, causal, sl)
    rp = None
    if rope:
        cos, sin = ops.rope_table(L + 32, d, 500000.0, DEV)
        rp = (cos, sin, torch.tensor([3, 0, 29][:B], dtype=torch.int32, device=DEV))
    res = {}
    for var in (BASE, 41, 4):
        dqkv = torch.full_like(qkv, float("nan"))
        ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, causal, sl, dqkv[:, :nq], dqkv[:, nq:nq + nk], dqkv[:, nq + nk:], rope=rp, variant=var)
        torch.cuda.synchronize()
        res[var] = dqkv
    tag = f"B{B} L{L} H{Hq}/{Hkv} causal={int(causal)} seqlens={seqlens} rope={int(rope)}"
    ok = True
    refs = None
    if not rope and B * L * Hq <= 2 * 2048 * 8:
        rq, rk, rv = ref_grads(q2.float().view(B, L, Hq, d), k2.float().view(B, L, Hkv, d), v2.float().view(B, L, Hkv, d), do.float().view(B, L, Hq, d), seqlens, causal)
        refs = torch.cat([rq.reshape(B * L, -1), rk.reshape(B * L, -1), rv.reshape(B * L, -1)], 1)
    for var in (41, 4):
        x = res[var]
        fin = bool(torch.isfinite(x.float()).all())
        parts = {"dq": slice(0, nq), "dk": slice(nq, nq + nk), "dv": slice(nq + nk, ld)}
        msg = f"[{tag}] variant {var}: finite={fin}"
        good = fin
        for name, sl_ in parts.items():
            e3 = float((x[:, sl_].float() - res[BASE][:, sl_].float()).abs().max())
            scale_ = float(res[BASE][:, sl_].float().abs().max())
            msg += f" {name}: |x - attn3|max={e3:.2e} (max |x| {scale_:.2e})"
            if refs is not None:
                er = float((x[:, sl_].float() - refs[:, sl_]).abs().max())
                e3r = float((res[BASE][:, sl_].float() - refs[:, sl_]).abs().max())
                msg += f" vs fp32 {er:.2e} (attn3 {e3r:.2e})"
                good &= er <= max(2.0 * e3r, 2e-2 * max(scale_, 1.0))
            else:
                good &= e3 <= 3e-2 * max(scale_, 1.0)
        print(msg + ("  OK" if good else "  **MISMATCH**"), flush=True)
        if not good:
            ok = False
            base = refs if refs is not None else res[BASE].float()
            diff = torch.nan_to_num((x.float() - base).abs(), nan=1e9).view(B, L, ld)
            for name, sl_ in parts.items():
                dd = diff[:, :, sl_]
                per_row = dd.amax(dim=2)[0]
                if L % 32 == 0:
                    print(f"   {name} per-32-row segment max error (sample 0):", [f"{v:.1e}" for v in per_row.view(-1, 32).amax(-1).tolist()][:64])
                print(f"   {name} per-16-column max error:", [f"{v:.1e}" for v in dd.amax(dim=(0, 1)).view(-1, 16).amax(-1).tolist()][:32])
    same = torch.equal(res[4], res[41])
    print(f"[{tag}] placed streams == serialised streams bit for bit: {same}", flush=True)
    return ok and same


def timeit(fn, it=5):
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


def bench(B, L, Hq, Hkv, causal=True, variants=(0, 4)):
    d = 128
    ld = (Hq + 2 * Hkv) * d
    nq, nk = Hq * d, Hkv * d
    qkv = (torch.randn(B * L, ld, device=DEV) * 0.5).bfloat16()
    q2, k2, v2 = qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:]
    o, lse = ops.attn_fwd(q2, k2, v2, B, L, Hq, Hkv, d, d ** -0.5, causal, None)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    fl = 2.5 * 4.0 * B * Hq * L * L * d / (2 if causal else 1)
    for var in [BASE if v == 0 else v for v in variants]:
        ms = timeit(lambda: ops.attn_bwd(q2, k2, v2, o, do, lse, B, L, Hq, Hkv, d, d ** -0.5, causal, None, dqkv[:, :nq], dqkv[:, nq:nq + nk], dqkv[:, nq + nk:], variant=var))
        print(f"[bench bwd B{B} L{L} H{Hq}/{Hkv} causal={int(causal)}] variant {var}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (all of the backward incl. delta)", flush=True)


if __name__ == "__main__":
    BASE = _base_variant()
    print(f"comparison kernels: variant {BASE}", flush=True)
    quick = "--quick" in sys.argv
    cases = [  # B, L, Hq, Hkv, causal, seqlens
        (1, 64, 2, 1, True, None),
        (1, 128, 2, 1, True, None),
        (1, 256, 2, 2, True, None),
        (2, 200, 4, 2, True, [200, 137]),
        (2, 333, 8, 2, True, [333, 256]),
        (1, 513, 4, 2, True, None),
        (2, 200, 2, 2, False, [200, 77]),
        (2, 256, 4, 1, True, [1, 256]),
        (1, 300, 6, 3, True, None),
        (2, 384, 12, 4, True, [300, 384]),
        (2, 320, 16, 2, True, [320, 191]),
        (1, 1024, 4, 2, True, None),
        (2, 2048, 8, 2, True, [2048, 1715]),
        (1, 2048, 4, 4, False, None),
    ]
    allok = True
    for c in cases:
        try:
            allok &= run_case(*c)
        except Exception as e:
            print(f"[{c}] EXCEPTION {type(e).__name__}: {e}", flush=True)
            allok = False
    for c in ((2, 333, 8, 2, True, [333, 256]), (1, 512, 4, 2, True, None)) if BASE == 3 else ():     # (the generic kernels have no fused inverse RoPE)
        try:
            allok &= run_case(*c, rope=True)
        except Exception as e:
            print(f"[{c} rope] EXCEPTION {type(e).__name__}: {e}", flush=True)
            allok = False
    print("ALL CASES OK" if allok else "SOME CASES FAILED", flush=True)
    if not quick:
        bench(4, 2048, 32, 8)
        bench(16, 2048, 32, 8)
        bench(16, 2048, 32, 8)
        bench(8, 4096, 32, 8)
