"""Test infrastructure: a CPU model of the ARITHMETIC of the d == 128 attention forward stream (metamorph_amd/csrc/attn4.hip,
tools/gen_attn4.py) and the adversarial score distributions the hostile-input parity tests run on.

The model restates, in fp32 torch, exactly what the kernel computes and in which precision -- not how it schedules it:

  * the score MFMAs take q and k as stored: s' = q . k - m is an fp32 sum of exact bf16 x bf16 products (units of the RAW dot product);
    the softmax scale enters in fp32, P = exp2(fp32(c * s')) with c = fp32(scale) * fp32(log2 e)  (v_pk_mul_f32 + v_exp_f32)
  * per wave (64 query rows of a 256-row block) and 64-key tile: m is the STALE running row maximum; tile 0 sets m = row maximum;
    afterwards m moves only when some row of the wave grew by more than 2^THR over it, i.e. c * growth > THR ("deferred rescale": the
    branch of gen_attn4.emit_decide), and then for every row of the wave by max(growth, 0)
  * row sums take the fp32 P, bf16(P) feeds O += P V (fp32 accumulate); o = bf16(O / l), lse = (log2 l + c * m) ln 2
  * only a wave's last tile is masked (causal diagonal / the sample's length); K / V rows beyond the sample arrive as zeros

`prescale="bf16"` gives the arithmetic of rounds 1-4 instead (Q~ = bf16(q * c) as the MFMA operand, the chain in the log2 domain): the
tests use it to show what the extra rounding cost (error versus max |s|).

It returns o, lse and the per-wave count of deferred-rescale branches, which the GPU test compares with the kernel's own tally
(mm355_attn_fwd_debug): "the branch ran as often as its decision rule says" is then a measured fact.  Differences between model and kernel
are accumulation order (MFMA vs torch matmul) and v_exp_f32's last bit.

`flash_bf16_backward` is the yardstick of the backward tests: the textbook flash-attention backward in the reference stack's precision
(bf16 P / dS / O operands, fp32 accumulation and softmax statistics: what torch SDPA's fused kernels compute at the reference's call site,
metamorph_llama.py:349-359), evaluated from fp32 scores of the un-rescaled bf16 q.
"""
import math

import torch

LOG2E = 1.4426950408889634
THR = 6.0
D = 128


def f32(x):
    return torch.tensor(x, dtype=torch.float32)


def sl2_of(scale):
    """c = scale * log2 e exactly as the kernels form it (fp32 product of two fp32 values)"""
    return f32(scale) * f32(LOG2E)


def attn4_forward_model(q, k, v, seqlens, causal, scale, prescale="fp32"):
    """q [B, L, Hq, 128], k / v [B, L, Hkv, 128] bf16 (CPU); seqlens list[int] | None.
    -> o [B, L, Hq, 128] bf16, lse [B, Hq, L] fp32, counts int32 [B, Hq, ceil(L / 256), 4]."""
    B, L, Hq, d = q.shape
    assert d == D
    Hkv = k.shape[2]
    rep = Hq // Hkv
    nblk = (L + 255) // 256
    sl2 = sl2_of(scale)
    if prescale == "bf16":                                       # rounds 1-4: the scale folded into a re-rounded bf16 copy of q
        qt = (q.float() * sl2).to(torch.bfloat16).float()
        c = f32(1.0)
    else:
        qt = q.float()
        c = sl2
    thr = f32(THR) / c
    kf, vf = k.float(), v.float()
    o = torch.zeros(B, L, Hq, d, dtype=torch.bfloat16)
    lse = torch.zeros(B, Hq, L, dtype=torch.float32)
    counts = torch.zeros(B, Hq, nblk, 4, dtype=torch.int32)
    ninf = float("-inf")
    for b in range(B):
        seqlen = L if seqlens is None else min(int(seqlens[b]), L)
        for hq in range(Hq):
            hk = hq // rep
            K = kf[b, :, hk].clone()
            V = vf[b, :, hk].clone()
            K[seqlen:] = 0                                       # rows beyond the sample: out of range of the buffer descriptor -> zeros
            V[seqlen:] = 0
            for xb in range(nblk):
                q0 = xb * 256
                if q0 >= seqlen:
                    continue                                     # whole block is padding: zeros
                kv_end = min(seqlen, q0 + 256) if causal else seqlen
                T = (kv_end + 63) >> 6
                for w in range(4):
                    qw0 = q0 + w * 64
                    if qw0 >= seqlen:
                        continue                                 # tw = -1: the wave only moves tiles
                    tw = min(T - 1, min(qw0 + 63, seqlen - 1) >> 6) if causal else T - 1
                    rows = torch.clamp(torch.arange(qw0, qw0 + 64), max=L - 1)          # loads clamp to the last row of the sample slab
                    Q = qt[b, rows, hq]                                                   # [64, d]
                    qg = torch.arange(qw0, qw0 + 64)
                    lim = (torch.clamp(qg, max=seqlen - 1) if causal else torch.full((64,), seqlen - 1)) - tw * 64
                    m = torch.zeros(64)
                    l = torch.zeros(64)
                    O = torch.zeros(64, d)
                    n = 0
                    for t in range(tw + 1):
                        Kt = torch.zeros(64, d)
                        Vt = torch.zeros(64, d)
                        hi_ = min(L, t * 64 + 64)
                        Kt[:hi_ - t * 64] = K[t * 64:hi_]
                        Vt[:hi_ - t * 64] = V[t * 64:hi_]
                        s = Q @ Kt.t()                                                    # [64 q, 64 keys], raw q . k units
                        if t == tw:
                            s = s.masked_fill(torch.arange(64)[None, :] > lim[:, None], ninf)
                        if t == 0:
                            rm = s.max(dim=1).values
                            rm = torch.where(rm == ninf, torch.zeros(()), rm)
                            m = rm.clone()
                            sp = s - m[:, None]
                        else:
                            sp = s - m[:, None]
                            rm = sp.max(dim=1).values
                            if bool((rm > thr).any()):
                                n += 1
                                delta = torch.clamp(rm, min=0.0)
                                al = torch.exp2(-delta * c)
                                O = O * al[:, None]
                                l = l * al
                                sp = sp - delta[:, None]
                                m = m + delta
                        p = torch.exp2(sp * c)
                        l = l + p.sum(dim=1)
                        O = O + p.to(torch.bfloat16).float() @ Vt
                    valid = qg < seqlen
                    inv = torch.where(valid & (l > 0), 1.0 / l, torch.zeros(()))
                    in_L = qg < L
                    o[b, qg[in_L], hq] = (O * inv[:, None]).to(torch.bfloat16)[in_L]
                    lw = torch.where(valid, (torch.log2(l) + m * c) * 0.6931471805599453, torch.zeros(()))
                    lse[b, hq, qg[in_L]] = lw[in_L]
                    counts[b, hq, xb, w] = n
    return o, lse, counts


def last_tile_of_wave(L, seqlen, causal):
    """tw per (block, wave) as the kernel computes it; -1 = the wave has no valid row.  [ceil(L / 256), 4] int"""
    nblk = (L + 255) // 256
    out = torch.full((nblk, 4), -1, dtype=torch.int32)
    for xb in range(nblk):
        q0 = xb * 256
        if q0 >= seqlen:
            continue
        kv_end = min(seqlen, q0 + 256) if causal else seqlen
        T = (kv_end + 63) >> 6
        for w in range(4):
            qw0 = q0 + w * 64
            if qw0 < seqlen:
                out[xb, w] = min(T - 1, min(qw0 + 63, seqlen - 1) >> 6) if causal else T - 1
    return out


def fp32_scores(q, k, b, hq, hk, scale):
    return (q[b, :, hq].float() @ k[b, :, hk].float().t()) * scale


# ------------------------------------------------------------------------------------------------ adversarial inputs

def _base(B, L, Hq, Hkv, seed, sigma):
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(B, L, Hq, D, generator=g) * sigma).to(torch.bfloat16)
    k = (torch.randn(B, L, Hkv, D, generator=g) * sigma).to(torch.bfloat16)
    v = torch.randn(B, L, Hkv, D, generator=g).to(torch.bfloat16)
    return q, k, v


def hostile_inputs(kind, B, L, Hq, Hkv, seed=0):
    """-> q [B, L, Hq, 128], k, v [B, L, Hkv, 128] bf16 whose scaled scores (scale = 128^-1/2) follow the named adversarial pattern.
    The structured part of a score rides on ONE coordinate: q[..., c] = A, k[..., c] = g(key) -> s += scale * A * g(key)."""
    scale = D ** -0.5
    A = 8.0
    unit = scale * A                                             # natural-log score units per unit of k[..., c]
    if kind == "rising":
        # every 64-key tile raises every row's maximum by 8 / unit * unit * log2 e = 8.16 log2 units: the branch fires at EVERY tile
        q, k, v = _base(B, L, Hq, Hkv, seed, 0.5)
        q[..., 0] = A
        tile = torch.arange(L) // 64
        k[..., 0] = (8.0 * tile.float())[None, :, None].to(torch.bfloat16)
        return q, k, v
    if kind == "one_row":
        # benign scores, except ONE query row per 64-row wave (row 17) that meets keys carrying +12 log2 units from key tile 3 on: the
        # wave-wide branch fires because of a single row; every other row of the wave gets its own (tiny, or exactly 1) factor
        q, k, v = _base(B, L, Hq, Hkv, seed, 0.5)
        q[..., 1] = 0
        q[:, 17::64, :, 1] = A
        k[..., 1] = 0
        late = torch.zeros(L)
        late[200::37] = 12.0 / (unit * LOG2E)
        k[..., 1] = late[None, :, None].to(torch.bfloat16)
        return q, k, v
    if kind == "sink":
        # attention sink: key 0 at +30 for every row, the rest N(0, 4): the first tile's maximum is never exceeded (stale maximum 43 log2
        # units above everything that follows: P down to 2^-60), except for rows that meet a second sink at key 300 (+45): one big rescale
        q, k, v = _base(B, L, Hq, Hkv, seed, 2.0)
        q[..., 0] = A
        k[..., 0] = 0
        k[:, 0, :, :] = 0
        k[:, 0, :, 0] = 30.0 / unit
        if L > 300:
            k[:, 300, :, :] = 0
            k[:, 300, :, 0] = 45.0 / unit
        return q, k, v
    if kind == "cliff":
        # every score -40 except one key at +40 late in the sequence: the running maximum jumps by 115 log2 units (factor 2^-115 on O and l);
        # a second cliff of +110 later: the factor exp2(-101 log2 e ...) underflows fp32 to ZERO, O and l restart from the new tile alone
        q, k, v = _base(B, L, Hq, Hkv, seed, 0.25)
        q[..., 0] = A
        k[..., 0] = -40.0 / unit
        k[:, (L * 5) // 8, :, 0] = 40.0 / unit
        k[:, (L * 7) // 8, :, 0] = 150.0 / unit
        return q, k, v
    if kind == "threshold":
        # growth just below / at the nearest round value below / just above the branch threshold 6 / (scale * log2 e) = 47.05239 raw units, in
        # samples 0 / 1 / 2: ONE non-zero coordinate (q0, k0), every other zero, so s' = q0 * k0 is exact in fp32 (bf16 x bf16): 47.05078
        # (threshold - 0.0016), 47.0 and 47.0625 (threshold + 0.0101).  The condition is "growth > threshold": only sample 2 takes the branch;
        # samples 0 / 1 carry P = 2^5.9998 / 2^5.9933 against the stale maximum
        assert B == 3
        q = torch.zeros(B, L, Hq, D, dtype=torch.bfloat16)
        k = torch.zeros(B, L, Hkv, D, dtype=torch.bfloat16)
        g = torch.Generator().manual_seed(seed)
        v = torch.randn(B, L, Hkv, D, generator=g).to(torch.bfloat16)
        for b, (q0, k0) in enumerate(((1.140625, 41.25), (1.0, 47.0), (1.5, 31.375))):
            q[b, :, :, 0] = q0
            k[b, 70::64, :, 0] = k0                               # one such key in every tile from tile 1 on: growth happens once, at tile 1
        return q, k, v
    if kind == "benign":
        return _base(B, L, Hq, Hkv, seed, 0.7)
    if kind == "wide":
        # trained-checkpoint-like: scores N(0, 9^2) (|s| to ~ 45 over a 2048 x 2048 head) with a few retrieval keys at +35 on top
        q, k, v = _base(B, L, Hq, Hkv, seed, 3.0)
        g = torch.Generator().manual_seed(seed + 7)
        q[..., 3] = A
        k[..., 3] = 0
        idx = torch.randint(0, L, (max(2, L // 100),), generator=g)
        k[:, idx, :, 3] = 35.0 / unit
        return q, k, v
    raise ValueError(kind)


# ------------------------------------------------------------------------------------------------ backward yardstick

def flash_bf16_forward(q, k, v, seqlens, causal, scale):
    """the forward half of flash_bf16_backward alone: o [B, L, Hq, d] bf16"""
    return flash_bf16_backward(q, k, v, None, seqlens, causal, scale)[0]


def flash_bf16_backward(q, k, v, do, seqlens, causal, scale):
    """Textbook flash-attention forward + backward of ONE batch in the reference stack's precision (what torch SDPA's fused bf16 kernels
    compute at metamorph_llama.py:349-359): scores from the bf16 q / k in fp32, softmax statistics fp32, P and dS rounded to bf16 before the
    matmuls that consume them, O rounded to bf16 before delta = sum(dO * O); fp32 accumulation everywhere.
    q [B, L, Hq, d], k / v [B, L, Hkv, d], do [B, L, Hq, d] bf16 -> (o bf16, dq, dk, dv fp32 in the same layouts)."""
    B, L, Hq, d = q.shape
    Hkv = k.shape[2]
    rep = Hq // Hkv
    bf = torch.bfloat16
    o = torch.zeros(B, L, Hq, d, dtype=bf)
    dq = torch.zeros(B, L, Hq, d)
    dk = torch.zeros(B, L, Hkv, d)
    dv = torch.zeros(B, L, Hkv, d)
    for b in range(B):
        n = L if seqlens is None else min(int(seqlens[b]), L)
        if n == 0:
            continue
        for hq in range(Hq):
            hk = hq // rep
            Q, K, V = q[b, :n, hq].float(), k[b, :n, hk].float(), v[b, :n, hk].float()
            s = (Q @ K.t()) * scale
            if causal:
                s = s.masked_fill(~torch.ones(n, n, dtype=torch.bool).tril(), float("-inf"))
            lse_ = torch.logsumexp(s, -1)
            p = torch.exp(s - lse_[:, None])
            pb = p.to(bf).float()
            ob = (pb @ V).to(bf)
            o[b, :n, hq] = ob
            if do is None:
                continue
            dO = do[b, :n, hq].float()
            delta = (dO * ob.float()).sum(-1)
            dp = dO @ V.t()
            ds = (p * (dp - delta[:, None])).to(bf).float()
            dq[b, :n, hq] = (ds @ K) * scale
            dk[b, :n, hk] += (ds.t() @ Q) * scale
            dv[b, :n, hk] += pb.t() @ dO
    return o, dq, dk, dv


# ------------------------------------------------------------------------------------------------ the backward streams' own arithmetic

def attn4_backward_model(q, k, v, o, do, lse, seqlens, causal, scale, noise=None):
    """CPU model of the ARITHMETIC of the d == 128 backward streams (metamorph_amd/csrc/attn4_bwd.hip, tools/gen_attn4_bwd.py; round 6:
    the hostile backward cases now assert "at most two bf16 steps from this model" instead of absolute floors around a yardstick).

    What the two kernels compute, and in which precision (not how they schedule it):
      * delta_i = sum_d bf16(dO) * bf16(O)  in fp32  (mm355_attn_bwd_prep; O = the forward kernel's bf16 output)
      * the score chains start from fp32 C operands: X = -lse_i * fp32(1 / scale) + q_i . k_j and Y = -delta_i + dO_i . v_j -- fp32 sums of
        exact bf16 x bf16 products on q, k, v, dO AS STORED (nstat_kernel / the dQ kernel's constant tuples; `lse` = the forward kernel's)
      * P = exp2(c * X), c = fp32(scale) * fp32(log2 e) (v_mul_f32 + v_exp_f32); masked scores (causal, keys >= the sample's length) are -inf;
        dS = P * Y in fp32
      * bf16(P) feeds dV^T += dO^T P, bf16(dS) feeds dK^T += Q^T dS and dQ^T += K^T dS^T (fp32 accumulation; the GQA group sum happens in
        the dK / dV accumulators); the softmax scale is applied ONCE to the finished dK / dQ accumulators, then one rounding to bf16
      * query rows >= the sample's length produce dq = 0 (`keep`), key rows >= it dk = dv = 0
    Both kernels recompute the SAME fp32 X, so one P / dS serves all three gradients here.  Differences between model and kernel:
    accumulation order inside the MFMAs and v_exp_f32's last bit.

    noise = a seed: every fp32 ADDEND of the two score chains (the C operand and the 128-term dot product) carries independent relative
    noise of 2^-21 -- the size of the rounding differences between two fp32 summation orders of the same chain (the MFMAs accumulate 8
    k-steps onto C; this model adds one torch matmul to C).  Where the chains cancel (dP - delta under a dominant sink; X = s - lse at
    |s| ~ 10^2-10^3 raw units feeding exp2) that noise is amplified into dS and through the key sum into dq / dk: the spread between noisy
    runs and the plain run is the CONDITIONING of each output, which the GPU test adds to its two-bf16-step bar.

    q [B, L, Hq, 128], k / v [B, L, Hkv, 128], o / do [B, L, Hq, 128] bf16, lse [B, Hq, L] fp32 (CPU) -> dq, dk, dv bf16 in the same layouts."""
    gen = None if noise is None else torch.Generator().manual_seed(int(noise))

    def jitter(t):
        return t if gen is None else t * (1.0 + (torch.rand(t.shape, generator=gen) * 2.0 - 1.0) * 2.0 ** -21)

    B, L, Hq, d = q.shape
    assert d == D
    Hkv = k.shape[2]
    rep = Hq // Hkv
    c = sl2_of(scale)
    inv_scale = f32(1.0) / f32(scale)
    bf = torch.bfloat16
    dq = torch.zeros(B, L, Hq, d, dtype=bf)
    dk = torch.zeros(B, L, Hkv, d, dtype=bf)
    dv = torch.zeros(B, L, Hkv, d, dtype=bf)
    for b in range(B):
        n = L if seqlens is None else min(int(seqlens[b]), L)
        if n == 0:
            continue
        for hk in range(Hkv):
            K, V = k[b, :n, hk].float(), v[b, :n, hk].float()
            dk_acc = torch.zeros(n, d)
            dv_acc = torch.zeros(n, d)
            for hq in range(hk * rep, (hk + 1) * rep):
                Q, dO, O = q[b, :n, hq].float(), do[b, :n, hq].float(), o[b, :n, hq].float()
                delta = (dO * O).sum(-1)
                X = jitter((-lse[b, hq, :n] * inv_scale)[:, None].expand(n, n)) + jitter(Q @ K.t())
                Y = jitter(-delta[:, None].expand(n, n)) + jitter(dO @ V.t())
                P = torch.exp2(c * X)
                if causal:
                    P = P.masked_fill(~torch.ones(n, n, dtype=torch.bool).tril(), 0.0)
                dS = P * Y
                Pb, dSb = P.to(bf).float(), dS.to(bf).float()
                dv_acc += Pb.t() @ dO
                dk_acc += dSb.t() @ Q
                dq[b, :n, hq] = ((dSb @ K) * f32(scale)).to(bf)
            dk[b, :n, hk] = (dk_acc * f32(scale)).to(bf)
            dv[b, :n, hk] = dv_acc.to(bf)
    return dq, dk, dv
