"""torch.autograd glue around the libmm355 kernels.

Each Function's forward AND backward are sequences of HIP kernels (metamorph_amd.ops); autograd is used
only to order them and to carry the activation gradients between the coarse blocks.  Weight gradients are
written by the backward kernels straight into persistent gradient buffers (`param.grad` becomes a view of
them) -- the "contiguous_gradients" behaviour of the reference's DeepSpeed ZeRO-2 config
(reference scripts/zero2.json:23) -- so no per-step gradient allocation or extra accumulate pass exists.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.autograd import Function

from . import ops

BF16 = torch.bfloat16
# Kernel-composition switches of the decoder layer.  The product runs the defaults (each one measured the winner on MI355X, DESIGN.md
# section 4); tools/ and tests flip them through set_variant() for same-box A/B runs -- there is no environment switch.
#   dw_tn           weight / input gradients on the operands as they lie in memory (contraction-major ping-pong GEMM, no transposed copies):
#                   TN 1.04-1.10 PFLOP/s, NN 1.25-1.30 vs 1.45-1.5 for the row-major kernel + explicit transposes -> off
#   dw_pair         down_proj and qkv weight gradients as ONE launch when that saves a wave of workgroups (896 + 384 tiles = 5 waves, not 4 + 2)
#   norm_t          backward rebuilds the RMSNorm outputs contraction-major from the saved rstd in one pass (rmsnorm_apply_t)
#   fuse_swiglu     SwiGLU in the gate|up GEMM's epilogue (same bits, one pass less)
#   fuse_rope       RoPE in the q|k|v GEMM's epilogue / inverse RoPE in the attention backward epilogues (same bits)
#   fuse_swiglu_bwd SwiGLU backward in the down_proj input-gradient GEMM's epilogue (same bits)
#   decode_graph    the per-token decode step is captured once as a hipGraph and replayed
VARIANTS = {"dw_tn": False, "dw_pair": True, "norm_t": True, "fuse_swiglu": True, "fuse_rope": True, "fuse_swiglu_bwd": True, "decode_graph": True, "decode_fused": True,
            # keep the transposed copies of the decoder weights (the B operands of the input-gradient GEMMs) from one backward call to
            # the next until the optimizer rewrites the parameters: under gradient accumulation the four transposes per layer are made
            # once per optimizer step instead of once per micro-batch (+2 bytes per decoder parameter while a window is open)
            "wt_cache": False,
            "decode_fold_rows": 4,                           # decode step: fold the RMSNorms into the GEMVs' operand reads up to this many rows
            # one launch for an input-gradient GEMM and the weight-gradient GEMM that reads the same dy: (dn2, dW_gate_up) and (dX_o, dW_o)
            # -- the mechanism of dw_pair (mm355_gemm_pair_bf16: same kernel body, same bits), saving one ramp and tail per pair
            "dx_pair": True,
            # prompt pass of a cached decode: q|k|v, o and down projections through the split-K GEMM (a few hundred rows: the plain kernels are a
            # latency chain over K there)
            "prefill_splitk": True,
            # decode step of MORE than 16 sequences: one pass over the weights through the split-K GEMM instead of one GEMV pass per 16 rows
            "decode_wide_gemm": True,
            # ... with RoPE + cache append, the two RMSNorms and SwiGLU folded into the reduce launches of the split projections (nine launches per
            # layer instead of thirteen; same bits)
            "decode_wide_fused": True, "decode_wide_gu_gemv": True,
            # the prompt pass of ONE sequence (a few hundred rows): the same reduce launches (RoPE + the K / V rows straight into the cache, both
            # RMSNorms), attention reading K / V from the cache rows -- same bits as the prefill_splitk pass, five launches per layer less
            "prefill_fused": True}


def set_variant(name, value):
    """Flip one composition switch (tools / tests); returns the previous value."""
    if name not in VARIANTS:
        raise KeyError(f"unknown variant {name!r}: {sorted(VARIANTS)}")
    old, VARIANTS[name] = VARIANTS[name], (int(value) if isinstance(VARIANTS[name], int) and not isinstance(VARIANTS[name], bool) else bool(value))
    return old


_CUS = 256                                                   # MI355X: one 256x256 tile per CU at a time


# ------------------------------------------------------------------------------------------------
# parameter / gradient storage helpers
# ------------------------------------------------------------------------------------------------

def _adjacent(ts):
    """True if the 2-D contiguous tensors `ts` are back-to-back rows of ONE storage object (tensors that merely
    happen to sit next to each other in the caching allocator do not count: a view over them would outgrow the
    first tensor's storage)."""
    t0 = ts[0]
    ptr = t0.data_ptr()
    st = t0.untyped_storage()
    total = sum(t.numel() * t.element_size() for t in ts)
    if ptr - st.data_ptr() + total > st.nbytes():
        return False
    for t in ts:
        if (not t.is_contiguous() or t.data_ptr() != ptr or t.shape[1:] != t0.shape[1:] or t.dtype != t0.dtype
                or t.untyped_storage().data_ptr() != st.data_ptr()):
            return False
        ptr += t.numel() * t.element_size()
    return True


def _view_rows(first, rows):
    """A [rows, cols] tensor aliasing the storage that starts at `first` (a contiguous 2-D tensor)."""
    out = first.new_empty(0)
    out.set_(first.untyped_storage(), first.storage_offset(), (rows, first.shape[1]), (first.shape[1], 1))
    return out


def fused_weight(params):
    """One [sum rows, K] tensor holding `params` back to back; re-points the parameters into it if needed."""
    datas = [p.data for p in params]
    rows = sum(d.shape[0] for d in datas)
    if _adjacent(datas):
        return _view_rows(datas[0], rows)
    buf = torch.empty((rows, datas[0].shape[1]), device=datas[0].device, dtype=datas[0].dtype)
    off = 0
    for p, d in zip(params, datas):
        buf[off:off + d.shape[0]].copy_(d)
        p.data = buf[off:off + d.shape[0]]
        off += d.shape[0]
    return buf


# Bumped by every optimizer step that rewrites parameters through raw pointers (Zero2AdamW / Zero3AdamW: mm355_adamw_shard and
# in-place collectives do not touch torch's per-tensor version counters); derived operand caches key on it.
_PARAM_GENERATION = [0]


def bump_param_generation():
    _PARAM_GENERATION[0] += 1


def param_generation():
    return _PARAM_GENERATION[0]


_LAYER_GRAD_HOOK = None


def set_layer_grad_hook(fn):
    """fn(layer) is called at the end of every DecoderLayerFn.backward (Zero2AdamW starts that layer's gradient
    reduce-scatter there, overlapping it with the backward of the earlier layers); None removes it."""
    global _LAYER_GRAD_HOOK
    _LAYER_GRAD_HOOK = fn


_PARAM_READY_HOOK = None


def set_param_ready_hook(fn):
    """fn(key) is called before a group of trainable parameters is first read in a forward pass -- key = the decoder layer
    module, or None for everything outside the decoder layers.  Zero2AdamW's asynchronous update makes the compute stream
    wait there for that segment's update (and all-gather) instead of at the end of step()."""
    global _PARAM_READY_HOOK
    _PARAM_READY_HOOK = fn


def params_ready(key=None, backward=False):
    """key = the decoder layer module about to be read (None: everything outside the decoder layers); backward=True when the
    reader is that layer's backward pass (Zero3AdamW then also attaches the layer's gradient slot)."""
    if _PARAM_READY_HOOK is not None:
        _PARAM_READY_HOOK(key, backward)


def grad_target(p):
    """(buffer, accumulate) for the gradient of parameter `p`."""
    if p.grad is not None:
        return p.grad, True
    buf = getattr(p, "_mm_grad_buf", None)
    if buf is None or buf.shape != p.shape or buf.device != p.device or buf.dtype != p.dtype:
        buf = torch.empty_like(p.data)
        p._mm_grad_buf = buf
    return buf, False


def commit_grad(p, buf):
    if p.grad is None:
        p.grad = buf


def fused_grad_target(params):
    """(fused buffer [sum rows, K], accumulate) whose row blocks are / become the .grad of `params`."""
    grads = [p.grad for p in params]
    rows = sum(p.shape[0] for p in params)
    if all(g is not None for g in grads) and _adjacent(grads):
        return _view_rows(grads[0], rows), True, None
    if all(g is None for g in grads):
        bufs = []
        for p in params:
            b = getattr(p, "_mm_grad_buf", None)
            bufs.append(b if (b is not None and b.shape == p.shape and b.device == p.device) else None)
        if any(b is None for b in bufs) or not _adjacent(bufs):
            big = torch.empty((rows, params[0].shape[1]), device=params[0].device, dtype=params[0].dtype)
            off = 0
            bufs = []
            for p in params:
                p._mm_grad_buf = big[off:off + p.shape[0]]
                bufs.append(p._mm_grad_buf)
                off += p.shape[0]
        return _view_rows(bufs[0], rows), False, bufs
    # mixed state (some grads set, some not): compute into a scratch buffer, then add per parameter
    return torch.empty((rows, params[0].shape[1]), device=params[0].device, dtype=params[0].dtype), None, None


def commit_fused_grad(params, fused, accumulate, bufs):
    if accumulate is True:
        return
    if accumulate is False:
        for p, b in zip(params, bufs):
            if p.requires_grad:
                p.grad = b
        return
    off = 0
    for p in params:
        blk = fused[off:off + p.shape[0]]
        off += p.shape[0]
        if not p.requires_grad:
            continue
        tgt, acc = grad_target(p)
        ops.axpy_(tgt, blk.contiguous(), None, 1.0, acc)
        commit_grad(p, tgt)


def _padded_rows(R, long_k=False):
    """K dimension a transposed copy of R rows gets: a multiple of 8 (legal GEMM K), or of 128 for a long contraction headed for
    the ping-pong kernel (whole pairs of 64-wide K tiles: keeps a ragged token count off the register-staged fallback kernel)."""
    return (R + 127) // 128 * 128 if (long_k and R >= 512) else (R + 7) // 8 * 8


def transpose_padded(x2d, Rp=None):
    """x [R, C] -> [C, Rp] with zero padding columns (they add nothing to a product contracted over them); Rp defaults to R
    rounded up to 8."""
    R, C = x2d.shape
    Rp = _padded_rows(R) if Rp is None else Rp
    buf = torch.empty((C, Rp), device=x2d.device, dtype=BF16)
    if Rp != R:
        buf[:, R:].zero_()                                    # only the padding columns
    ops.transpose(x2d, out=buf[:, :R])
    return buf


def _dw_operands(dy2d, x2d, dyT=None, xT=None):
    """Contraction-major operands (dy^T [N, Mp], x^T [K, Mp]) of a weight-gradient GEMM in NT form; a copy a producer already
    wrote (dyT / xT) fixes Mp, otherwise long contractions are padded to whole pairs of K tiles."""
    R = (dy2d if dy2d is not None else x2d).shape[0]
    Rp = dyT.shape[1] if dyT is not None else (xT.shape[1] if xT is not None else _padded_rows(R, long_k=True))
    return (transpose_padded(dy2d, Rp) if dyT is None else dyT, transpose_padded(x2d, Rp) if xT is None else xT)


def weight_grad_gemm(dy2d, x2d, out, accumulate, dyT=None, xT=None):
    """out[N,K] (+)= dy[M,N]^T @ x[M,K].  Token counts that are whole pairs of 64-row tiles go straight through the
    contraction-major ping-pong kernel (operands as they lie in memory, fragments gathered by ds_read_b64_tr_b16); ragged
    or small problems fall back to explicit transposes.  Opt-in (MM355VARIANTS["dw_tn"]=1): see the note at the top."""
    if dyT is None and xT is None and VARIANTS["dw_tn"]:
        big = ((dy2d.shape[1] + 255) // 256) * ((x2d.shape[1] + 255) // 256) >= 128
        if big and ops.gemm_tn_supported(dy2d, x2d) and ops.gemm_pp_operands_ok(dy2d.shape[0], dy2d, x2d):
            ops.gemm_tn(dy2d, x2d, out, accumulate=accumulate)
            return
    # dyT / xT: contraction-major copies a producer already wrote (the row-major argument may then be None)
    a, b = _dw_operands(dy2d, x2d, dyT, xT)
    ops.gemm(a, b, out=out, accumulate=accumulate)


def _pair_saves_a_wave(rows0, cols0, rows1, cols1, contraction):
    """Two weight gradients [rows, cols] over `contraction` token rows: does one paired launch of the 256x256 kernel need fewer
    waves of workgroups than two launches?  (Both must be problems the ping-pong kernel takes on its own: >= 200 tiles.)"""
    t0 = -(-rows0 // 256) * -(-cols0 // 256)
    t1 = -(-rows1 // 256) * -(-cols1 // 256)
    kp = _padded_rows(contraction, long_k=True)
    if min(t0, t1) < 200 or kp % 128:
        return False
    return -(-(t0 + t1) // _CUS) < -(-t0 // _CUS) + -(-t1 // _CUS)


_WT_CACHE = {}                                               # (storage, offset, shape, strides) -> ((generation, source versions...), transposed copy)


def transposed_weight(w, sources=None):
    """w [N, K] -> [K, Np] (transpose_padded).  With VARIANTS["wt_cache"] AND `sources` given the copy is kept until the parameters change:
    same bits, one transpose per optimizer step.  `sources` = the parameters w is (a fused view of); it is passed only at the decoder
    layer's own call sites, whose operands are long-lived parameters or views of the optimizer's flat buffer -- never by the generic
    input_grad_gemm (LinearFn / LinearWBFn also see per-forward temporaries, whose address a later tensor can reuse).  An entry is valid
    for one parameter generation (`bump_param_generation`: every optimizer step / checkpoint load, which write behind torch's back) AND one
    set of torch version counters of the sources (any in-place write through torch -- copy_, load_state_dict, mul_ -- invalidates it
    without anybody having to bump)."""
    if sources is None or not VARIANTS["wt_cache"]:
        return transpose_padded(w)
    key = (w.untyped_storage().data_ptr(), w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    stamp = (param_generation(),) + tuple(p._version for p in sources)
    hit = _WT_CACHE.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    if _WT_CACHE and next(iter(_WT_CACHE.values()))[0][0] != stamp[0]:
        _WT_CACHE.clear()                                     # a new generation: every cached copy is stale
    wt = transpose_padded(w)
    _WT_CACHE[key] = (stamp, wt)
    return wt


def drop_transposed_weights():
    _WT_CACHE.clear()


def _pairable(a0, b0, a1, b1):
    """Two plain NT problems worth one launch: each big enough for the 256 x 256 kernel on its own (>= 200 tiles), the first a whole
    number of 8 workgroups (the second problem's workgroups then keep their XCD), both eligible for mm355_gemm_pair_bf16."""
    t0 = -(-a0.shape[0] // 256) * -(-b0.shape[0] // 256)
    t1 = -(-a1.shape[0] // 256) * -(-b1.shape[0] // 256)
    return min(t0, t1) >= 200 and t0 % 8 == 0 and ops.gemm_pair_supported(a0, b0, a1, b1)


def input_grad_gemm(dy2d, w, out=None, residual=None):
    """dx[M,K] = dy[M,N] @ w[N,K]  (+ residual); the weight is read untransposed whenever the ping-pong kernel applies"""
    if VARIANTS["dw_tn"] and ops.gemm_nn_supported(dy2d, w):
        return ops.gemm_nn(dy2d, w, out=out, residual=residual)
    return ops.gemm(dy2d, transposed_weight(w), out=out, residual=residual)


# ------------------------------------------------------------------------------------------------
# LLaMA decoder layer as ONE autograd node
# ------------------------------------------------------------------------------------------------

class LayerMeta:
    """Geometry shared by all decoder layers of one forward pass."""

    def __init__(self, B, L, Hq, Hkv, d, I, eps, cos, sin, seqlens, recompute=False, pos_offset=None, c2p=None, p2c=None):
        self.B, self.L, self.Hq, self.Hkv, self.d, self.I, self.eps = B, L, Hq, Hkv, d, I, eps
        self.cos, self.sin, self.seqlens = cos, sin, seqlens
        self.pos_offset = pos_offset        # int32 [B] or None: RoPE position of row (b, l) = l + pos_offset[b] (left-padded batches)
        # padding-free rows (ragged batches): the decoder's row-wise work (norms, GEMMs, SwiGLU, residuals) runs on the COMPACT layout --
        # the sum(len) valid rows back to back, rounded up to 256 with zero rows -- and only q|k|v -> attention -> o visits the padded
        # [B, L] layout the attention kernels address (per-sample lengths from row b * L).  c2p int32 [rows]: padded row of every compact
        # row (-1: a zero row of the tail); p2c int32 [B * L]: compact row of every padded row (-1: padding).  None: x IS the padded layout.
        self.c2p, self.p2c = c2p, p2c
        # prompt pass of a cached decode (decoder_prefill sets it): a few hundred rows -- the q|k|v, o and down projections take the split-K
        # GEMM (ops.gemm_splitk); inference only
        self.prompt_pass = False
        self.scale = d ** -0.5
        # gradient checkpointing (reference train.py:1443-1449 + `--gradient_checkpointing True` in every launch script): keep only
        # the layer input, re-run the layer's forward kernels at the start of its backward
        self.recompute = recompute


PROMPT_GU_SPLITK_ROWS = 64


def decoder_layer_forward(x, layer, m: LayerMeta):
    """x [B*L, h] -> (y, saved tensors).  HF LlamaDecoderLayer semantics (reference call site
    metamorph_llama.py:349-359)."""
    att, mlp = layer.self_attn, layer.mlp
    wqkv = fused_weight([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
    wgu = fused_weight([mlp.gate_proj.weight, mlp.up_proj.weight])
    n1, rstd1 = ops.rmsnorm_fwd(x, layer.input_layernorm.weight, m.eps, want_rstd=True)
    if m.p2c is None and VARIANTS["fuse_rope"] and ops.gemm_rope_supported(n1, wqkv, m.Hq, m.Hkv, m.d, m.cos):
        qkv = ops.gemm_rope(n1, wqkv, m.B, m.L, m.Hq, m.Hkv, m.d, m.cos, m.sin, pos_offset=m.pos_offset)     # rotation in the GEMM epilogue: same bits
        del n1
    else:
        qkv = ops.gemm_splitk(n1, wqkv) if m.prompt_pass else ops.gemm(n1, wqkv)
        del n1
        if m.p2c is not None:                                  # compact rows -> the padded layout attention addresses (padding rows: zeros)
            qkv = ops.rows_gather(qkv, m.p2c)
        ops.rope_qk_(qkv, m.B, m.L, m.Hq, m.Hkv, m.d, m.cos, m.sin, pos_offset=m.pos_offset)
    nq, nk = m.Hq * m.d, m.Hkv * m.d
    o, lse = ops.attn_fwd(qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:], m.B, m.L, m.Hq, m.Hkv, m.d, m.scale, True, m.seqlens)
    if m.c2p is not None:
        o = ops.rows_gather(o, m.c2p)                          # back to compact rows for o_proj and everything after it
    x2 = ops.gemm_splitk(o, att.o_proj.weight, residual=x) if m.prompt_pass else ops.gemm(o, att.o_proj.weight, residual=x)
    n2, rstd2 = ops.rmsnorm_fwd(x2, layer.post_attention_layernorm.weight, m.eps, want_rstd=True)
    if VARIANTS["fuse_swiglu"] and ops.gemm_swiglu_supported(n2, wgu, m.I):
        gu, act = ops.gemm_swiglu(n2, wgu, m.I)                # SiLU(gate) * up formed in the GEMM epilogue: same bits, one pass less
        del n2
    else:
        # (prompt passes of up to 64 rows: gate|up through the split-K GEMM as well -- 5.65 -> 5.25 ms per 64-row pass; from 128 rows on the
        # plain 128 x 128 tiles are faster)
        gu = ops.gemm_splitk(n2, wgu) if (m.prompt_pass and n2.shape[0] <= PROMPT_GU_SPLITK_ROWS) else ops.gemm(n2, wgu)
        del n2
        act = ops.swiglu_fwd(gu, m.I)
    y = ops.gemm_splitk(act, mlp.down_proj.weight, residual=x2) if m.prompt_pass else ops.gemm(act, mlp.down_proj.weight, residual=x2)
    # rstd1 / rstd2 (fp32 [M] each): the backward pass rebuilds the TRANSPOSED norm outputs from them in one pass (rmsnorm_apply_t)
    return y, (qkv, o, lse, x2, gu, rstd1, rstd2)


def _rmsnorm_backward(dn, x, w, eps, dres):
    """dres + d rmsnorm / dx; the weight gradient (if trained) lands in the parameter's gradient buffer in the same launch pair."""
    if not w.requires_grad:
        return ops.rmsnorm_bwd(dn, x, w, eps, dres=dres, dw_f32=None)
    buf, acc = grad_target(w)
    dx = ops.rmsnorm_bwd_wgrad(dn, x, w, eps, buf, acc, dres=dres)
    commit_grad(w, buf)
    return dx


class DecoderLayerFn(Function):
    @staticmethod
    def forward(ctx, x, layer, meta, *weights):
        y, saved = decoder_layer_forward(x, layer, meta)
        ctx.layer, ctx.meta = layer, meta
        if meta.recompute:
            del saved
            ctx.save_for_backward(x)
        else:
            ctx.save_for_backward(x, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer, m = ctx.layer, ctx.meta
        params_ready(layer, backward=True)                                     # sharded parameters (ZeRO-3): gather before use
        if len(ctx.saved_tensors) == 1:                                        # checkpointed: same kernels, same inputs => same bits
            (x,) = ctx.saved_tensors
            _, (qkv, o, lse, x2, gu, rstd1, rstd2) = decoder_layer_forward(x, layer, m)
        else:
            x, qkv, o, lse, x2, gu, rstd1, rstd2 = ctx.saved_tensors
        att, mlp = layer.self_attn, layer.mlp
        dy = dy.contiguous()
        h = x.shape[1]
        nq, nk = m.Hq * m.d, m.Hkv * m.d
        dev = x.device

        # ---- MLP ----
        gu_params = [mlp.gate_proj.weight, mlp.up_proj.weight]
        # full fine-tune on whole 64-row tiles: SwiGLU backward writes act^T and dgu^T itself (no transpose passes over them)
        fused_t = (not VARIANTS["dw_tn"] and dy.shape[0] % 64 == 0 and m.I % 64 == 0 and mlp.down_proj.weight.requires_grad
                   and all(p.requires_grad for p in gu_params))
        wdT = wdT_made = None
        if fused_t and VARIANTS["fuse_swiglu_bwd"]:
            wdT = wdT_made = transposed_weight(mlp.down_proj.weight, sources=(mlp.down_proj.weight,))
            if not ops.gemm_swiglu_bwd_supported(dy, wdT, gu, m.I):
                wdT = None
        if wdT is not None:
            # d act = dy Wd formed, rounded and consumed in the GEMM's epilogue: same bits as the two launches, no d act round trip
            dgu, actT, dguT = ops.gemm_swiglu_bwd(dy, wdT, gu, m.I)
            act = None
            del wdT
        else:
            dact = ops.gemm(dy, wdT_made) if wdT_made is not None else input_grad_gemm(dy, mlp.down_proj.weight)   # [M, I]
            if fused_t:
                dgu, actT, dguT = ops.swiglu_bwd_t(gu, dact, m.I)
                act = None
            else:
                dgu, act = ops.swiglu_bwd(gu, dact, m.I, want_act=mlp.down_proj.weight.requires_grad)
                actT = dguT = None
            del dact
        qkv_params = [att.q_proj.weight, att.k_proj.weight, att.v_proj.weight]
        held = None                      # down_proj's weight-gradient problem, kept back to share a launch with qkv's
        if mlp.down_proj.weight.requires_grad:
            wd = mlp.down_proj.weight
            buf, acc = grad_target(wd)
            if (VARIANTS["dw_pair"] and not VARIANTS["dw_tn"] and all(p.requires_grad for p in qkv_params)
                    and _pair_saves_a_wave(wd.shape[0], wd.shape[1], sum(p.shape[0] for p in qkv_params), h, dy.shape[0])):
                held = _dw_operands(dy, act, xT=actT) + (buf, acc)
            else:
                weight_grad_gemm(dy, act, buf, acc, xT=actT)
            commit_grad(wd, buf)
        del act, actT
        wgu = fused_weight(gu_params)
        dn2 = None
        if any(p.requires_grad for p in gu_params):
            fb, acc, bufs = fused_grad_target(gu_params)
            if VARIANTS["dw_tn"] or not VARIANTS["norm_t"]:
                dn2 = input_grad_gemm(dgu, wgu)
                weight_grad_gemm(dgu, ops.rmsnorm_fwd(x2, layer.post_attention_layernorm.weight, m.eps), fb, bool(acc), dyT=dguT)
            else:                                                              # norm output, contraction-major, from the saved rstd
                rp = dguT.shape[1] if dguT is not None else _padded_rows(x2.shape[0], long_k=True)
                n2T = ops.rmsnorm_apply_t(x2, layer.post_attention_layernorm.weight, rstd2, rp)
                a0, b0 = _dw_operands(dgu, None, dyT=dguT, xT=n2T)
                wguT = transposed_weight(wgu, sources=gu_params)
                if VARIANTS["dx_pair"] and not VARIANTS["dw_tn"] and _pairable(a0, b0, dgu, wguT):
                    dn2 = torch.empty((dgu.shape[0], wguT.shape[0]), device=dev, dtype=BF16)
                    ops.gemm_pair(a0, b0, fb, bool(acc), dgu, wguT, dn2, False)   # the weight gradient's long-K tiles first
                else:
                    dn2 = ops.gemm(dgu, wguT)
                    ops.gemm(a0, b0, out=fb, accumulate=bool(acc))
                del n2T, a0, b0, wguT
            commit_fused_grad(gu_params, fb, acc, bufs)
        if dn2 is None:
            dn2 = input_grad_gemm(dgu, wgu)                                     # [M, h]
        del dgu, dguT
        dx2 = _rmsnorm_backward(dn2, x2, layer.post_attention_layernorm.weight, m.eps, dy)     # dy + d rmsnorm
        del dn2

        # ---- attention ----
        do = None
        if att.o_proj.weight.requires_grad:
            buf, acc = grad_target(att.o_proj.weight)
            if VARIANTS["dx_pair"] and not VARIANTS["dw_tn"]:
                a0, b0 = _dw_operands(dx2, o)
                woT = transposed_weight(att.o_proj.weight, sources=(att.o_proj.weight,))
                if _pairable(a0, b0, dx2, woT):
                    do = torch.empty((dx2.shape[0], woT.shape[0]), device=dev, dtype=BF16)
                    ops.gemm_pair(a0, b0, buf, acc, dx2, woT, do, False)
                else:
                    do = ops.gemm(dx2, woT)
                    ops.gemm(a0, b0, out=buf, accumulate=acc)
                del a0, b0, woT
            else:
                weight_grad_gemm(dx2, o, buf, acc)
            commit_grad(att.o_proj.weight, buf)
        if do is None:
            do = input_grad_gemm(dx2, att.o_proj.weight)                        # [M, Hq*d]
        dqkv = torch.empty_like(qkv)
        fuse_rope = VARIANTS["fuse_rope"] and ops.attn_bwd_rope_supported(m.d)      # inverse RoPE of dq / dk in the attention kernels' epilogues
        o_att = o
        if m.p2c is not None:                                  # padding-free rows: o / d o to the padded layout (qkv was saved in it)
            o_att, do = ops.rows_gather(o, m.p2c), ops.rows_gather(do, m.p2c)
        ops.attn_bwd(qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:], o_att, do, lse, m.B, m.L, m.Hq, m.Hkv, m.d,
                     m.scale, True, m.seqlens, dqkv[:, :nq], dqkv[:, nq:nq + nk], dqkv[:, nq + nk:],
                     rope=(m.cos, m.sin, m.pos_offset) if fuse_rope else None)
        del do, o_att
        if not fuse_rope:
            ops.rope_qk_(dqkv, m.B, m.L, m.Hq, m.Hkv, m.d, m.cos, m.sin, inverse=True, pos_offset=m.pos_offset)
        if m.c2p is not None:
            dqkv = ops.rows_gather(dqkv, m.c2p)                # gradients of the compact q|k|v rows (tail rows: zeros)
        wqkv = fused_weight(qkv_params)
        if VARIANTS["dw_tn"] and ops.gemm_nn_supported(dqkv, wqkv):
            dn1 = ops.gemm_nn(dqkv, wqkv)
        else:
            dn1 = ops.gemm(dqkv, transposed_weight(wqkv, sources=qkv_params))
        if any(p.requires_grad for p in qkv_params):
            fb, acc, bufs = fused_grad_target(qkv_params)
            if VARIANTS["dw_tn"] or not VARIANTS["norm_t"]:
                n1, n1T = ops.rmsnorm_fwd(x, layer.input_layernorm.weight, m.eps), None
            else:
                n1, n1T = None, ops.rmsnorm_apply_t(x, layer.input_layernorm.weight, rstd1, _padded_rows(x.shape[0], long_k=True))
            if held is not None:
                a1, b1 = _dw_operands(dqkv, n1, xT=n1T)
                if ops.gemm_pair_supported(held[0], held[1], a1, b1):
                    ops.gemm_pair(held[0], held[1], held[2], held[3], a1, b1, fb, bool(acc))
                else:                                                          # e.g. an operand beyond 2 GiB
                    ops.gemm(held[0], held[1], out=held[2], accumulate=held[3])
                    ops.gemm(a1, b1, out=fb, accumulate=bool(acc))
                held = None
                del a1, b1
            else:
                weight_grad_gemm(dqkv, n1, fb, bool(acc), xT=n1T)
            commit_fused_grad(qkv_params, fb, acc, bufs)
            del n1, n1T
        del dqkv
        dx = _rmsnorm_backward(dn1, x, layer.input_layernorm.weight, m.eps, dx2)
        if _LAYER_GRAD_HOOK is not None:                                       # every gradient of this layer is final now
            _LAYER_GRAD_HOOK(layer)
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def decoder_layer(x, layer, meta):
    params_ready(layer)
    ws = [p for p in layer.parameters()]
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in ws)):
        return DecoderLayerFn.apply(x, layer, meta, *ws)
    return decoder_layer_forward(x, layer, meta)[0]


# ------------------------------------------------------------------------------------------------
# small composable nodes: RMSNorm, Linear(+bias), GELU, row gather, cosine loss
# ------------------------------------------------------------------------------------------------

class RmsNormFn(Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return ops.rmsnorm_fwd(x, w.data if isinstance(w, torch.nn.Parameter) else w, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dw = torch.zeros(x.shape[-1], device=x.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        dx = ops.rmsnorm_bwd(dy.contiguous(), x, w, ctx.eps, dres=None, dw_f32=dw)
        dwb = None
        if dw is not None:
            dwb = torch.empty_like(w)
            ops.axpy_(dwb, dw, None, 1.0, False)
        return dx, dwb, None


class LinearFn(Function):
    """y = x W^T (+ b); weight / bias gradients go straight to the parameters' gradient buffers."""

    @staticmethod
    def forward(ctx, x, weight, bias, module):
        ctx.module = module
        ctx.save_for_backward(x)
        return ops.gemm(x, weight, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mod = ctx.module
        dy = dy.contiguous()
        w, b = mod.weight, mod.bias
        dx = input_grad_gemm(dy, w) if ctx.needs_input_grad[0] else None
        if w.requires_grad:
            buf, acc = grad_target(w)
            weight_grad_gemm(dy, x, buf, acc)
            commit_grad(w, buf)
        if b is not None and b.requires_grad:
            s = torch.zeros(b.shape[0], device=dy.device, dtype=torch.float32)
            ops.colsum_f32(dy, s)
            buf, acc = grad_target(b)
            ops.axpy_(buf, s, None, 1.0, acc)
            commit_grad(b, buf)
        return dx, None, None, None


def _param_grad_f32(p, s_f32):
    """p.grad (+)= s_f32 (an fp32 reduction result such as a bias / norm-weight gradient), into the parameter's gradient buffer."""
    buf, acc = grad_target(p)
    ops.axpy_(buf.view(-1), s_f32.view(-1), None, 1.0, acc)
    commit_grad(p, buf)


def _linear_grads(mod, dy2d, x2d):
    """Weight and bias gradients of y = x W^T + b from dy (may be a strided column block) and the saved input."""
    if mod.weight.requires_grad:
        buf, acc = grad_target(mod.weight)
        weight_grad_gemm(dy2d, x2d, buf, acc)
        commit_grad(mod.weight, buf)
    if mod.bias is not None and mod.bias.requires_grad:
        s = torch.zeros(mod.bias.shape[0], device=dy2d.device, dtype=torch.float32)
        ops.colsum_f32(dy2d, s)
        _param_grad_f32(mod.bias, s)


# ------------------------------------------------------------------------------------------------
# SigLIP encoder with a backward pass (freeze_vision=False, reference siglip_encoder.py:138-141 running HF
# SiglipEncoderLayer under grad).  Coarse nodes like DecoderLayerFn: forward keeps (x, qkv, o, lse, x2, fc1 pre-activation),
# backward recomputes the two LayerNorms and the GELU, parameter gradients go straight to their gradient buffers.
# ------------------------------------------------------------------------------------------------

class SiglipGeo:
    def __init__(self, N, P, heads, d, eps, recompute=False):
        self.N, self.P, self.heads, self.d, self.eps = N, P, heads, d, eps
        self.scale = d ** -0.5
        self.recompute = recompute


def _siglip_qkv(h1, att, hv):
    qkv = torch.empty((h1.shape[0], 3 * hv), device=h1.device, dtype=BF16)
    for i, proj in enumerate((att.q_proj, att.k_proj, att.v_proj)):     # HF stores k, v, q, out separately (with biases)
        ops.gemm(h1, proj.weight.data, out=qkv[:, i * hv:(i + 1) * hv], bias=proj.bias.data)
    return qkv


def siglip_layer_forward(x, layer, g):
    att, mlp = layer.self_attn, layer.mlp
    hv = x.shape[1]
    ln1, ln2 = layer.layer_norm1, layer.layer_norm2
    h1 = ops.layernorm_fwd(x, ln1.weight.data, ln1.bias.data, g.eps)
    qkv = _siglip_qkv(h1, att, hv)
    del h1
    o, lse = ops.attn_fwd(qkv[:, :hv], qkv[:, hv:2 * hv], qkv[:, 2 * hv:], g.N, g.P, g.heads, g.heads, g.d, g.scale, False, None)
    x2 = ops.gemm(o, att.out_proj.weight.data, bias=att.out_proj.bias.data, residual=x)
    h2 = ops.layernorm_fwd(x2, ln2.weight.data, ln2.bias.data, g.eps)
    f_pre = ops.gemm(h2, mlp.fc1.weight.data, bias=mlp.fc1.bias.data)
    del h2
    f = ops.gelu_fwd(f_pre, ops.GELU_TANH)
    y = ops.gemm(f, mlp.fc2.weight.data, bias=mlp.fc2.bias.data, residual=x2)
    return y, (qkv, o, lse, x2, f_pre)


class SiglipLayerFn(Function):
    @staticmethod
    def forward(ctx, x, layer, g, *params):
        y, saved = siglip_layer_forward(x, layer, g)
        ctx.layer, ctx.g = layer, g
        if g.recompute:
            del saved
            ctx.save_for_backward(x)
        else:
            ctx.save_for_backward(x, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer, g = ctx.layer, ctx.g
        if len(ctx.saved_tensors) == 1:
            (x,) = ctx.saved_tensors
            _, (qkv, o, lse, x2, f_pre) = siglip_layer_forward(x, layer, g)
        else:
            x, qkv, o, lse, x2, f_pre = ctx.saved_tensors
        att, mlp = layer.self_attn, layer.mlp
        ln1, ln2 = layer.layer_norm1, layer.layer_norm2
        hv = x.shape[1]
        dev = x.device
        dy = dy.contiguous()
        # ---- MLP
        f = ops.gelu_fwd(f_pre, ops.GELU_TANH)
        df = input_grad_gemm(dy, mlp.fc2.weight)
        _linear_grads(mlp.fc2, dy, f)
        del f
        dfp = ops.gelu_bwd(f_pre, df, ops.GELU_TANH)
        del df
        h2 = ops.layernorm_fwd(x2, ln2.weight.data, ln2.bias.data, g.eps)
        dh2 = input_grad_gemm(dfp, mlp.fc1.weight)
        _linear_grads(mlp.fc1, dfp, h2)
        del h2, dfp
        dw, db = torch.zeros(hv, device=dev, dtype=torch.float32), torch.zeros(hv, device=dev, dtype=torch.float32)
        dx2 = ops.layernorm_bwd(dh2, x2, ln2.weight.data, g.eps, dw, db, dres=dy)       # dy + d LayerNorm2
        del dh2
        if ln2.weight.requires_grad:
            _param_grad_f32(ln2.weight, dw)
            _param_grad_f32(ln2.bias, db)
        # ---- attention
        do = input_grad_gemm(dx2, att.out_proj.weight)
        _linear_grads(att.out_proj, dx2, o)
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv[:, :hv], qkv[:, hv:2 * hv], qkv[:, 2 * hv:], o, do, lse, g.N, g.P, g.heads, g.heads, g.d, g.scale, False, None,
                     dqkv[:, :hv], dqkv[:, hv:2 * hv], dqkv[:, 2 * hv:])
        del do
        h1 = ops.layernorm_fwd(x, ln1.weight.data, ln1.bias.data, g.eps)
        dh1 = None
        for i, proj in enumerate((att.q_proj, att.k_proj, att.v_proj)):
            blk = dqkv[:, i * hv:(i + 1) * hv]
            dh1 = input_grad_gemm(blk, proj.weight, residual=dh1)
            _linear_grads(proj, blk, h1)
        del h1, dqkv
        dw, db = torch.zeros(hv, device=dev, dtype=torch.float32), torch.zeros(hv, device=dev, dtype=torch.float32)
        dx = ops.layernorm_bwd(dh1, x, ln1.weight.data, g.eps, dw, db, dres=dx2)
        if ln1.weight.requires_grad:
            _param_grad_f32(ln1.weight, dw)
            _param_grad_f32(ln1.bias, db)
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class PatchEmbedFn(Function):
    """Conv2d(kernel = stride = patch) + bias + position embedding as im2col + GEMM; gradients of the three parameters."""

    @staticmethod
    def forward(ctx, images, emb, *params):
        w = emb.patch_embedding.weight
        k = w.shape[1] * w.shape[2] * w.shape[3]
        kp = (k + 7) // 8 * 8
        wpe = torch.zeros((w.shape[0], kp), device=w.device, dtype=BF16)
        wpe[:, :k].copy_(w.data.reshape(w.shape[0], k))
        p = w.shape[2]
        N, _, H, W = images.shape
        P = (H // p) * (W // p)
        cols = ops.im2col_patch(images, p, kp)
        x = ops.gemm(cols, wpe, bias=emb.patch_embedding.bias.data, residual=emb.position_embedding.weight.data, res_row_mod=P)
        ctx.emb, ctx.k, ctx.N, ctx.P = emb, k, N, P
        ctx.save_for_backward(cols)
        return x

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        emb, k, N, P = ctx.emb, ctx.k, ctx.N, ctx.P
        dy = dy.contiguous()
        w, b, pos = emb.patch_embedding.weight, emb.patch_embedding.bias, emb.position_embedding.weight
        hv = w.shape[0]
        if w.requires_grad:
            dwp = torch.empty((hv, cols.shape[1]), device=dy.device, dtype=BF16)
            weight_grad_gemm(dy, cols, dwp, False)
            buf, acc = grad_target(w)
            ops.axpy_(buf.view(hv, k), dwp[:, :k].contiguous(), None, 1.0, acc)
            commit_grad(w, buf)
        if b.requires_grad:
            s = torch.zeros(hv, device=dy.device, dtype=torch.float32)
            ops.colsum_f32(dy, s)
            _param_grad_f32(b, s)
        if pos.requires_grad:                                   # every image adds the same table: sum over the images
            s = torch.zeros(P * hv, device=dy.device, dtype=torch.float32)
            ops.colsum_f32(dy.view(N, P * hv), s)
            _param_grad_f32(pos, s)
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class BilinearL2NormFn(Function):
    @staticmethod
    def forward(ctx, x, side_in, side_out, normalize):
        ctx.args = (side_in, side_out, normalize)
        ctx.save_for_backward(x)
        return ops.bilinear_l2norm(x, side_in, side_out, normalize)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        side_in, side_out, normalize = ctx.args
        return ops.bilinear_l2norm_bwd(x, dy.contiguous(), side_in, side_out, normalize), None, None, None


def linear(x2d, module):
    if (x2d.shape[0] <= 32 and not (torch.is_grad_enabled() and (x2d.requires_grad or module.weight.requires_grad))
            and ops.gemv_supported(x2d, module.weight.data)):
        # decode shape: a handful of rows, inference only -> stream the weight once (mm355_gemv_bf16)
        return ops.gemv(x2d, module.weight.data, bias=None if module.bias is None else module.bias.data)
    return LinearFn.apply(x2d, module.weight, module.bias, module)


class KVCache:
    """Post-RoPE keys and values of every decoder layer for a BATCH of sequences: [layers, batch, max_len, Hkv*d] bf16 each (one sequence:
    batch = 1).  Every row of the batch has its own length; the write positions live on the device as well (pos_dev / len_dev, int32
    [batch]) so that a decode step -- ONE pass over the weights for all rows -- is replayable as a hipGraph."""

    def __init__(self, n_layers, max_len, width, device, Hq=None, d=None, batch=1):
        self.batch = batch
        self.k = torch.empty((n_layers, batch, max_len, width), device=device, dtype=BF16)
        self.v = torch.empty((n_layers, batch, max_len, width), device=device, dtype=BF16)
        self.max_len = max_len
        self.lengths = [0] * batch                                               # host mirror of pos_dev
        self.pos_dev = torch.zeros(batch, device=device, dtype=torch.int32)      # row the next token of sequence b is written to
        self.len_dev = torch.ones(batch, device=device, dtype=torch.int32)       # = pos + 1: rows visible to that token
        self.ws = None
        if Hq is not None:                                                       # (sized for the whole batch: beyond 16 rows one attention launch serves all of them)
            self.ws = torch.zeros(int(ops._L().mm355_attn_decode_ws_floats(batch, Hq, d, max_len)), device=device,
                                  dtype=torch.float32)                           # arrival counters start at 0

    @property
    def length(self):
        """rows of the longest sequence (= THE length when batch == 1)"""
        return max(self.lengths)

    @length.setter
    def length(self, n):
        self.lengths = [n] * self.batch

    def set_length(self, n, row=None):
        if row is None:
            self.set_lengths([n] * self.batch)
        else:
            ls = list(self.lengths)
            ls[row] = n
            self.set_lengths(ls)

    def set_lengths(self, lengths):
        assert len(lengths) == self.batch
        self.lengths = [int(n) for n in lengths]
        pos = torch.tensor(self.lengths, dtype=torch.int32)
        self.pos_dev.copy_(pos)
        self.len_dev.copy_(pos + 1)


def decoder_prefill(x, layers, meta, cache, row=0):
    """Prompt pass of a cached decode: the training-path layer forward, keeping each layer's post-RoPE k / v rows.
    x [meta.B * L, h] -> hidden rows (pre final norm).  meta.B == 1: the prompt of sequence `row`; meta.B == cache.batch: all sequences
    at once (prompts of ONE length -- the beams of a beam search, a batch without padding)."""
    nq, nk = meta.Hq * meta.d, meta.Hkv * meta.d
    B = meta.B
    L = x.shape[0] // B
    assert B == 1 or (B == cache.batch and row == 0)
    meta.prompt_pass = VARIANTS["prefill_splitk"] and not torch.is_grad_enabled() and x.shape[0] <= 4096
    ident = torch.arange(L, device=x.device, dtype=torch.int32) if B == 1 else None
    if (B == 1 and meta.prompt_pass and VARIANTS["prefill_fused"] and meta.d % 16 == 0 and meta.pos_offset is None and meta.p2c is None
            and ops.gemm_splitk_splits(L, (meta.Hq + 2 * meta.Hkv) * meta.d, x.shape[1])):
        x = _prefill_layers_fused(x, layers, meta, cache, row, ident)
        cache.set_length(L, row)
        return x
    for i, layer in enumerate(layers):
        params_ready(layer)
        x, saved = decoder_layer_forward(x, layer, meta)
        qkv = saved[0]
        if B == 1:
            # the k / v column blocks of the fused activation -> cache rows: a strided row copy with 16-byte vectors (mm355_rows_gather with the
            # identity map: ~3 us; ATen's strided bf16 copy took 28 us per call, 1.3 of the 10.8 ms of a 512-row prompt pass)
            ops.rows_gather(qkv[:, nq:nq + nk], ident, out=cache.k[i, row, :L])
            ops.rows_gather(qkv[:, nq + nk:], ident, out=cache.v[i, row, :L])
        else:
            cache.k[i, :, :L].copy_(qkv[:, nq:nq + nk].view(B, L, nk))
            cache.v[i, :, :L].copy_(qkv[:, nq + nk:].view(B, L, nk))
        del saved
    if B == 1:
        cache.set_length(L, row)
    else:
        cache.set_length(L)
    return x


def _prefill_layers_fused(x, layers, meta, cache, row, ident):
    """decoder_prefill for one sequence of a few hundred rows, with what follows a split projection in its reduce launch: q|k|v -> RoPE at the
    row's position -> rotated k and v straight into the cache rows (mm355_gemm_splitk_rope_append_bf16 with a batch stride of 0: every row of
    the call belongs to the same sequence), attention reads K / V from the cache rows, o / down + residual + the RMSNorm that follows
    (mm355_gemm_splitk_norm_bf16).  The bits of decoder_layer_forward's prompt pass (rope_qk_ and rope_kv_append share their arithmetic)."""
    L = x.shape[0]
    nq = meta.Hq * meta.d
    n1 = None
    for i, layer in enumerate(layers):
        params_ready(layer)
        att, mlp = layer.self_attn, layer.mlp
        wqkv = fused_weight([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
        wgu = fused_weight([mlp.gate_proj.weight, mlp.up_proj.weight])
        if n1 is None:
            n1 = ops.rmsnorm_fwd(x, layer.input_layernorm.weight, meta.eps)
        kc, vc = cache.k[i, row], cache.v[i, row]                             # [max_len, width]
        rows_k = kc.as_strided((L, kc.shape[0], kc.shape[1]), (0, kc.stride(0), 1))
        rows_v = vc.as_strided((L, vc.shape[0], vc.shape[1]), (0, vc.stride(0), 1))
        qkv = ops.gemm_splitk_rope_append(n1, wqkv, meta.Hq, meta.Hkv, meta.d, meta.cos, meta.sin, ident, rows_k, rows_v)
        o, _ = ops.attn_fwd(qkv[:, :nq], kc[:L], vc[:L], 1, L, meta.Hq, meta.Hkv, meta.d, meta.scale, True, meta.seqlens)
        x2, n2 = ops.gemm_splitk_norm(o, att.o_proj.weight, layer.post_attention_layernorm.weight, meta.eps, residual=x)
        if VARIANTS["fuse_swiglu"] and ops.gemm_swiglu_supported(n2, wgu, meta.I):
            _, act = ops.gemm_swiglu(n2, wgu, meta.I)
        elif L <= PROMPT_GU_SPLITK_ROWS:
            act = ops.gemm_splitk_swiglu(n2, wgu, meta.I)
        else:
            act = ops.swiglu_fwd(ops.gemm(n2, wgu), meta.I)
        if i + 1 < len(layers):
            params_ready(layers[i + 1])
            x, n1 = ops.gemm_splitk_norm(act, mlp.down_proj.weight, layers[i + 1].input_layernorm.weight, meta.eps, residual=x2)
        else:
            x = ops.gemm_splitk(act, mlp.down_proj.weight, residual=x2)
    return x


def _decode_rows16(x, layers, meta, cos, sin, k, v, pos_dev, len_dev, ws, kv_bound):
    """<= 16 new rows (one per sequence) through every decoder layer against their cache rows k / v [layers, rows, max_len, width]:
    every weight is streamed ONCE for all rows (mm355_gemv* take M <= 16: up to four rows on the vector ALU, 5 .. 16 on MFMA), attention per
    row at its own length."""
    nq = meta.Hq * meta.d
    for i, layer in enumerate(layers):
        params_ready(layer)
        att, mlp = layer.self_attn, layer.mlp
        wqkv = fused_weight([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
        wgu = fused_weight([mlp.gate_proj.weight, mlp.up_proj.weight])
        if VARIANTS["decode_fused"] and meta.I % 2 == 0 and meta.d % 4 == 0:
            # five launches per layer: RMSNorm folded into the q|k|v and gate|up GEMVs' operand reads, RoPE + cache append and SwiGLU into
            # their epilogues, the flash-decoding merge into the chunk that finishes last (same bits as the nine-launch sequence below).
            # Five rows and more (the MFMA GEMVs): the norm runs as its own launch -- folded in, every workgroup would normalise ALL rows again
            # (measured, cached step of 32 layers: four rows 3.50 -> 3.38 ms with the norms folded, eight rows slower) -- seven launches, same bits.
            fold = x.shape[0] <= VARIANTS["decode_fold_rows"]
            n1 = x if fold else ops.rmsnorm_fwd(x, layer.input_layernorm.weight, meta.eps)
            qkv = ops.gemv_rope_append(n1, wqkv, meta.Hq, meta.Hkv, meta.d, cos, sin, pos_dev, k[i], v[i],
                                       norm_w=layer.input_layernorm.weight if fold else None, eps=meta.eps)
            o = ops.attn_decode(qkv[:, :nq], k[i], v[i], len_dev, kv_bound, meta.Hq, meta.Hkv, meta.d, meta.scale, workspace=ws)
            x2 = ops.gemv(o, att.o_proj.weight, residual=x)
            n2 = x2 if fold else ops.rmsnorm_fwd(x2, layer.post_attention_layernorm.weight, meta.eps)
            act = ops.gemv_swiglu(n2, wgu, meta.I, norm_w=layer.post_attention_layernorm.weight if fold else None, eps=meta.eps)
            x = ops.gemv(act, mlp.down_proj.weight, residual=x2)
            continue
        n1 = ops.rmsnorm_fwd(x, layer.input_layernorm.weight, meta.eps)
        qkv = ops.gemv(n1, wqkv)
        ops.rope_kv_append_(qkv, meta.Hq, meta.Hkv, meta.d, cos, sin, pos_dev, k[i], v[i])
        o = ops.attn_decode(qkv[:, :nq], k[i], v[i], len_dev, kv_bound, meta.Hq, meta.Hkv, meta.d, meta.scale, workspace=ws)
        x2 = ops.gemv(o, att.o_proj.weight, residual=x)
        n2 = ops.rmsnorm_fwd(x2, layer.post_attention_layernorm.weight, meta.eps)
        gu = ops.gemv(n2, wgu)
        act = ops.swiglu_fwd(gu, meta.I)
        x = ops.gemv(act, mlp.down_proj.weight, residual=x2)
    return x


def _decode_rows_gemm(x, layers, meta, cos, sin, k, v, pos_dev, len_dev, ws, kv_bound):
    """17 new rows and more (one per sequence) through every decoder layer in ONE pass over the weights (round 6; rounds 5: chunks of 16 rows,
    i.e. the 15 GB read once per chunk).  The GEMV kernels hold one 16-row MFMA operand; beyond it the projections take the split-K GEMM of
    the prompt pass (mm355_gemm_splitk_bf16: 64 x 128 tiles x K slices -- at 32 rows a weight-streaming problem with ~3 workgroups per CU),
    attention per row at its own length; RoPE + cache append, the two RMSNorms and SwiGLU ride in the reduce launches of the split projections
    (VARIANTS["decode_wide_fused"]; as launches of their own -- thirteen per layer instead of nine -- when off: same bits)."""
    nq = meta.Hq * meta.d
    if VARIANTS["decode_wide_fused"] and meta.d % 16 == 0:
        # what follows a split projection rides in its reduce launch (mm355_gemm_splitk_{rope_append,norm,swiglu}_bf16): 9 launches per layer
        n1 = None
        gu_gemv = VARIANTS["decode_wide_gu_gemv"] and x.shape[0] <= 32 and meta.I % 2 == 0 and ops.gemv_rows32_units(meta.I // 2)
        for i, layer in enumerate(layers):
            params_ready(layer)
            att, mlp = layer.self_attn, layer.mlp
            wqkv = fused_weight([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
            wgu = fused_weight([mlp.gate_proj.weight, mlp.up_proj.weight])
            if n1 is None:
                n1 = ops.rmsnorm_fwd(x, layer.input_layernorm.weight, meta.eps)
            qkv = ops.gemm_splitk_rope_append(n1, wqkv, meta.Hq, meta.Hkv, meta.d, cos, sin, pos_dev, k[i], v[i])
            o = ops.attn_decode(qkv[:, :nq], k[i], v[i], len_dev, kv_bound, meta.Hq, meta.Hkv, meta.d, meta.scale, workspace=ws)
            x2, n2 = ops.gemm_splitk_norm(o, att.o_proj.weight, layer.post_attention_layernorm.weight, meta.eps, residual=x)
            if gu_gemv:
                act = ops.gemv_swiglu(n2, wgu, meta.I)        # up to 32 rows: the weight stream of the 16-row kernel, a second x row group on its fragments
            else:
                act = ops.gemm_splitk_swiglu(n2, wgu, meta.I)
            if i + 1 < len(layers):
                params_ready(layers[i + 1])                   # (its input norm weight is read by this layer's last launch)
                x, n1 = ops.gemm_splitk_norm(act, mlp.down_proj.weight, layers[i + 1].input_layernorm.weight, meta.eps, residual=x2)
            else:
                x = ops.gemm_splitk(act, mlp.down_proj.weight, residual=x2)
        return x
    for i, layer in enumerate(layers):
        params_ready(layer)
        att, mlp = layer.self_attn, layer.mlp
        wqkv = fused_weight([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
        wgu = fused_weight([mlp.gate_proj.weight, mlp.up_proj.weight])
        n1 = ops.rmsnorm_fwd(x, layer.input_layernorm.weight, meta.eps)
        qkv = ops.gemm_splitk(n1, wqkv)
        ops.rope_kv_append_(qkv, meta.Hq, meta.Hkv, meta.d, cos, sin, pos_dev, k[i], v[i])
        o = ops.attn_decode(qkv[:, :nq], k[i], v[i], len_dev, kv_bound, meta.Hq, meta.Hkv, meta.d, meta.scale, workspace=ws)
        x2 = ops.gemm_splitk(o, att.o_proj.weight, residual=x)
        n2 = ops.rmsnorm_fwd(x2, layer.post_attention_layernorm.weight, meta.eps)
        act = ops.swiglu_fwd(ops.gemm_splitk(n2, wgu), meta.I)
        x = ops.gemm_splitk(act, mlp.down_proj.weight, residual=x2)
    return x


SHORT_KV = 1024                                              # mm355_attn_decode: a bound of <= 1024 rows is one key group (no merge, more workgroups)


def decode_kv_bound(cache):
    """The static upper bound on the cached lengths handed to mm355_attn_decode for the NEXT step: 1024 while every sequence (incl. the row
    being appended) still fits one key group -- the kernel then runs its single-group form --, the cache's capacity afterwards.  The host
    knows the lengths (cache.lengths); the bound is a launch parameter, so a captured step exists once per bound (DecodeStepGraph)."""
    return SHORT_KV if (cache.length + 1 <= SHORT_KV and cache.max_len >= SHORT_KV) else cache.max_len


def decoder_decode_row(x, layers, meta, cache, cos, sin, kv_bound=None):
    """One new row PER SEQUENCE against the cache (reference semantics: HF LlamaDecoderLayer with past_key_values, the whole batch in one
    forward per step -- metamorph_llama.py:711-717; the reference's own greedy loop recomputes the prefix instead, :502-597).
    x [batch, h] -> [batch, h]; row b is appended at cache.lengths[b].  The batch goes through the layers in ONE pass (16 rows at a time):
    the 15 GB of LLaMA-3-8B weights are read once per step, not once per sequence.  Every position-dependent input is read from device
    memory (cache.pos_dev / len_dev): the launch sequence is identical for every token, i.e. capturable once and replayable
    (DecodeStepGraph)."""
    if cache.length >= cache.max_len:
        raise ValueError(f"KV cache full ({cache.max_len} rows)")
    B = x.shape[0]
    if B != cache.batch:
        raise ValueError(f"{B} rows for a cache of {cache.batch} sequences")
    bound = decode_kv_bound(cache) if kv_bound is None else kv_bound
    if cache.length + 1 > bound:
        # the bound is a launch parameter (number of key groups): a sequence longer than it would silently attend to its first `bound` keys
        raise ValueError(f"decode bound {bound} is below the longest sequence + 1 ({cache.length + 1}); lengths change only through set_lengths")
    if B <= 16:
        y = _decode_rows16(x, layers, meta, cos, sin, cache.k, cache.v, cache.pos_dev, cache.len_dev, cache.ws, bound)
    elif VARIANTS["decode_wide_gemm"]:
        y = _decode_rows_gemm(x, layers, meta, cos, sin, cache.k, cache.v, cache.pos_dev, cache.len_dev, cache.ws, bound)
    else:                                                    # (rounds 5: one pass over the weights per chunk of 16 rows)
        y = torch.cat([_decode_rows16(x[c:c + 16], layers, meta, cos, sin, cache.k[:, c:c + 16], cache.v[:, c:c + 16], cache.pos_dev[c:c + 16],
                                      cache.len_dev[c:c + 16], cache.ws, bound) for c in range(0, B, 16)], 0)
    cache.pos_dev.add_(1)
    cache.len_dev.add_(1)
    cache.lengths = [n + 1 for n in cache.lengths]
    return y


class DecodeStepGraph:
    """decoder_decode_row captured as a hipGraph (5 - 7 launches x layers per step collapse into one graph launch; the per-token step is
    launch-bound otherwise) and replayed per token: copy the new rows (one per sequence) into `x_in`, replay, read `x_out`.  One capture
    per attention bound (decode_kv_bound: the short, single-key-group form while every sequence holds <= 1024 rows, the general one
    afterwards), made when first needed.  Falls back to eager launches if capture is not possible (functional.set_variant("decode_graph",
    False) forces that)."""

    def __init__(self, layers, meta, cache, cos, sin, h, device):
        self.args = (layers, meta, cache, cos, sin)
        self.cache = cache
        self.x_in = torch.zeros((cache.batch, h), device=device, dtype=BF16)
        self.graphs = {}                                     # kv bound -> (graph, x_out)
        self.capture_ok = VARIANTS["decode_graph"]
        if self.capture_ok:
            self._capture(decode_kv_bound(cache))

    @property
    def graph(self):                                         # (tools / tests: "is the step a graph?")
        return next(iter(self.graphs.values()))[0] if self.graphs else None

    def _capture(self, bound):
        cache = self.cache
        keep = list(cache.lengths)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up outside capture (writes a scratch row, undone below)
                decoder_decode_row(self.x_in, *self.args, kv_bound=bound)
            torch.cuda.current_stream().wait_stream(side)
            cache.set_lengths(keep)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                x_out = decoder_decode_row(self.x_in, *self.args, kv_bound=bound)
            cache.lengths = keep                                 # capture records, it does not run: host mirror unchanged
            self.graphs[bound] = (g, x_out)
        except Exception as e:                                   # pragma: no cover - depends on the runtime
            import warnings
            warnings.warn(f"hipGraph capture of the decode step failed ({e!r}); using eager launches")
            self.capture_ok = False
            self.graphs = {}
            cache.set_lengths(keep)

    def step(self, rows):
        """rows [batch, h] (one new row per sequence) -> their hidden rows [batch, h] (pre final norm; overwritten by the next step)"""
        if self.capture_ok:
            bound = decode_kv_bound(self.cache)
            if bound not in self.graphs:
                self._capture(bound)
        if not self.capture_ok:
            return decoder_decode_row(rows.contiguous().view(self.cache.batch, -1), *self.args)
        if self.cache.length >= self.cache.max_len:
            raise ValueError(f"KV cache full ({self.cache.max_len} rows)")
        g, x_out = self.graphs[bound]
        self.x_in.copy_(rows.view(self.cache.batch, -1))
        g.replay()
        self.cache.lengths = [n + 1 for n in self.cache.lengths]
        return x_out


class GeluFn(Function):
    @staticmethod
    def forward(ctx, x, kind):
        ctx.save_for_backward(x)
        ctx.kind = kind
        return ops.gelu_fwd(x, kind)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x, dy.contiguous(), ctx.kind), None


class TransposeFn(Function):
    """x [R, C] -> x^T [C, Rp] (zero-padded columns when Rp > R); backward transposes the live part back."""

    @staticmethod
    def forward(ctx, x, Rp):
        ctx.R = x.shape[0]
        return transpose_padded(x, Rp) if Rp is not None else _transpose_exact(x)

    @staticmethod
    def backward(ctx, dy):
        return _transpose_exact(dy[:, :ctx.R]), None


def _transpose_exact(x2d):
    R, C = x2d.shape
    out = torch.empty((C, R), device=x2d.device, dtype=BF16)
    ops.transpose(x2d, out=out)
    return out


class PadFn(Function):
    """t [N, K] (or [N]) -> zero-padded [Np, Kp] (or [Np]): legal GEMM operand shapes for odd widths; backward slices."""

    @staticmethod
    def forward(ctx, t, Np, Kp):
        ctx.shape = tuple(t.shape)
        src = t.data if isinstance(t, torch.nn.Parameter) else t
        if t.dim() == 1:
            out = torch.zeros((Np,), device=t.device, dtype=t.dtype)
            out[:t.shape[0]].copy_(src)
        else:
            out = torch.zeros((Np, Kp), device=t.device, dtype=t.dtype)
            out[:t.shape[0], :t.shape[1]].copy_(src)
        return out

    @staticmethod
    def backward(ctx, dy):
        sl = dy[:ctx.shape[0]] if len(ctx.shape) == 1 else dy[:ctx.shape[0], :ctx.shape[1]]
        return sl.contiguous(), None, None


class LinearWBFn(Function):
    """y = x W^T + b with weight / bias passed as TENSORS (gradients returned through autograd, e.g. to PadColsFn)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.gemm(x, weight, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = input_grad_gemm(dy, w) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            weight_grad_gemm(dy, x, dw, False)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            s = torch.zeros(dy.shape[1], device=dy.device, dtype=torch.float32)
            ops.colsum_f32(dy, s)
            db = torch.empty(dy.shape[1], device=dy.device, dtype=BF16)
            ops.axpy_(db, s, None, 1.0, False)
        return dx, dw, db


class RowsPermuteFn(Function):
    """out[r] = x[fwd[r]] (fwd[r] = -1 -> zero row); backward gathers with the inverse map.  Moves a left-padded batch to the
    right-padded row layout the attention kernels take (per-sample lengths from row 0) and back."""

    @staticmethod
    def forward(ctx, x, fwd_i32, inv_i32):
        ctx.save_for_backward(inv_i32)
        return ops.rows_gather(x, fwd_i32)

    @staticmethod
    def backward(ctx, dy):
        (inv,) = ctx.saved_tensors
        return ops.rows_gather(dy.contiguous(), inv), None, None


class RowsGatherFn(Function):
    """out[r] = x[idx[r]] (idx unique, >= 0); backward scatters into a zero tensor."""

    @staticmethod
    def forward(ctx, x, idx_i32):
        ctx.save_for_backward(idx_i32)
        ctx.shape = x.shape
        return ops.rows_gather(x, idx_i32)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, device=dy.device, dtype=BF16)
        ops.rows_scatter_add_(dx, dy.contiguous(), idx)
        return dx, None


class CosineLossFn(Function):
    """-mean_r cos(target_r, normalize(pred_r))  (metamorph_llama.py:433-435,449-455)."""

    @staticmethod
    def forward(ctx, pred_raw, target, normalize):
        cos_sum, dpred = ops.cosine_loss(pred_raw, target, normalize, want_grad=True)
        ctx.save_for_backward(dpred)
        R = pred_raw.shape[0]
        # scalar plumbing on a 1-element tensor
        return (cos_sum * (-1.0 / R)).reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return _scaled_copy(dpred, g), None, None


def _scaled_copy(saved, g):
    """saved * g into a FRESH tensor (a second backward through the same graph must not see a pre-scaled gradient)."""
    out = torch.empty_like(saved)
    ops.axpy_(out, saved, g.reshape(1).float().contiguous(), 1.0, False)
    return out


class MeanAbsLossFn(Function):
    """mean |target - pred|: the reference's `mse_loss_fn` (metamorph_llama.py:211-219, reached at :459 when neither
    normalize_vision nor apply_softmax is set -- the constructor default)."""

    @staticmethod
    def forward(ctx, pred, target, denom_rows=None):
        """denom_rows: the row count the reference divides by (len(target), :217) when it differs from the rows compared."""
        abs_sum, dpred = ops.mean_abs_loss(pred, target, want_grad=True)
        ctx.save_for_backward(dpred)
        R = pred.shape[0]
        ctx.scale = 1.0 if denom_rows in (None, R) else R / float(denom_rows)       # the kernel's gradient carries 1 / (R C)
        return (abs_sum * (ctx.scale / pred.numel())).reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return _scaled_copy(dpred, g * ctx.scale if ctx.scale != 1.0 else g), None, None


class SoftCELossFn(Function):
    """-(target * log(softmax(u / 0.07) + 1e-10)).sum(1).mean(), u = F.normalize(pred) when normalize_vision
    (metamorph_llama.py:434-447)."""

    @staticmethod
    def forward(ctx, pred_raw, target, normalize):
        loss_sum, dpred = ops.soft_ce_loss(pred_raw, target, normalize, 0.07, want_grad=True)
        ctx.save_for_backward(dpred)
        return (loss_sum * (1.0 / pred_raw.shape[0])).reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return _scaled_copy(dpred, g), None, None


class SoftmaxRowsFn(Function):
    """softmax(x / 0.07) over the feature dimension (tower side, siglip_encoder.py:210-211)."""

    @staticmethod
    def forward(ctx, x2d, temperature):
        y = ops.softmax_rows(x2d, temperature)
        ctx.save_for_backward(y)
        ctx.temperature = temperature
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.softmax_rows_bwd(y, dy.contiguous(), ctx.temperature), None


# ------------------------------------------------------------------------------------------------
# splice (K6): embedding gather + image rows + zero padding
# ------------------------------------------------------------------------------------------------

class SpliceFn(Function):
    @staticmethod
    def forward(ctx, embed_weight, proj2d, embed_module, plan_dev):
        ctx.embed_module, ctx.plan = embed_module, plan_dev
        ctx.proj_shape = None if proj2d is None else proj2d.shape
        h = embed_weight.shape[1]
        return ops.splice_gather(embed_weight, proj2d, plan_dev["src"], h)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        plan, emb = ctx.plan, ctx.embed_module
        dproj = None
        if ctx.proj_shape is not None and ctx.needs_input_grad[1]:
            dproj = ops.rows_gather(dout, plan["feat_row"][: ctx.proj_shape[0]])
        w = emb.weight
        if w.requires_grad and plan["emb_tok"].numel() > 0:
            buf, acc = grad_target(w)
            if not acc:
                buf.zero_()                       # rows of tokens that do not occur keep a zero gradient
            ops.embed_grad_(buf, dout, plan["emb_tok"], plan["emb_seg"], plan["emb_pos"], True)
            commit_grad(w, buf)
        return None, dproj, None, None


class EmbeddingFn(Function):
    """Plain `embed_tokens(ids)` WITH a gradient: the text-only training path (forward(images=None): the reference's splice returns early,
    metamorph_arch.py:184-191, and HF's LlamaModel embeds the ids itself).  Backward = the splice's deterministic segmented row sum
    (`mm355_embed_grad`: rows sorted by token id on the host, from the ids' host mirror if their mover registered one -- hostmirror.py --,
    else one D2H copy; this is not the hot path)."""

    @staticmethod
    def forward(ctx, embed_weight, embed_module, ids):
        ctx.embed_module = embed_module
        ctx.ids = ids
        flat = ids.reshape(-1).to(torch.int32)
        out = ops.splice_gather(embed_weight, None, flat, embed_weight.shape[1])
        return out.view(*ids.shape, embed_weight.shape[1])

    @staticmethod
    def backward(ctx, dout):
        w = ctx.embed_module.weight
        if w.requires_grad:
            from .hostmirror import host_array
            flat = host_array(ctx.ids).reshape(-1).astype(np.int64)
            order = np.argsort(flat, kind="stable")
            sorted_tok = flat[order]
            starts = np.flatnonzero(np.concatenate(([True], sorted_tok[1:] != sorted_tok[:-1])))
            dev = dout.device
            tok = torch.from_numpy(sorted_tok[starts].astype(np.int32)).to(dev)
            seg = torch.from_numpy(np.concatenate((starts, [sorted_tok.size])).astype(np.int32)).to(dev)
            pos = torch.from_numpy(order.astype(np.int32)).to(dev)
            buf, acc = grad_target(w)
            if not acc:
                buf.zero_()
            ops.embed_grad_(buf, dout.reshape(-1, dout.shape[-1]).contiguous(), tok, seg, pos, True)
            commit_grad(w, buf)
        return None, None, None


# ------------------------------------------------------------------------------------------------
# lm_head + shifted cross entropy without materialising [M, V] logits (K13)
# ------------------------------------------------------------------------------------------------

CE_CHUNK = 8192


class LinearCrossEntropyFn(Function):
    """loss = mean over rows with a target of CE(hidden_r @ W^T, target_r) -- ONE library call (mm355_linear_ce, include/mm355.h).

    Rows without a target were already dropped by the host plan (`ce_rows`), so no flop is spent on
    ignored positions.  Inside the library, per chunk of 8192 rows: logits GEMM (bf16) -> fp32 softmax / NLL kernel that overwrites
    the logits with d loss / d logits -> the two gradient GEMMs; the per-row NLL values are summed in a fixed order (bit-reproducible
    loss).  The [M,V] fp32 logits tensor of the reference (metamorph_llama.py:398-399) never exists.
    """

    @staticmethod
    def forward(ctx, hidden, weight, head_module, plan_dev, n_valid):
        loss, dhc, dw = ops.linear_ce(hidden, plan_dev["ce_rows"], plan_dev["ce_targets"], weight, need_dh=hidden.requires_grad,
                                      need_dw=weight.requires_grad, dw_f32=n_valid > CE_CHUNK)
        ctx.head_module, ctx.plan, ctx.hidden_shape = head_module, plan_dev, hidden.shape
        ctx.save_for_backward(dhc, dw)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dhc, dw = ctx.saved_tensors
        gs = g.reshape(1).float().contiguous()
        dh = None
        if dhc is not None:
            dh = ops.rows_gather(dhc, ctx.plan["ce_inv"])   # scatter back to [M, h] (zero rows elsewhere)
            ops.scale_(dh, gs, 1.0)
        w = ctx.head_module.weight
        if dw is not None and w.requires_grad:
            buf, acc = grad_target(w)
            ops.axpy_(buf, dw, gs, 1.0, acc)
            commit_grad(w, buf)
        return dh, None, None, None, None
