"""End-to-end CPU restatement of the hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

`forward(sd, cfg, batch)` reproduces, on a flat state dict with the reference's key names
(SURVEY.md section 8b), what `MetaMorphLlamaForCausalLM.forward` computes
(reference metamorph/model/language_model/metamorph_llama.py:603-660 -> :285-498), i.e.

  A3  SigLIP tower, hidden_states[-1] (pre post_layernorm)      siglip_encoder.py:138-213
  A4  mm_projector + detached regression targets                metamorph_arch.py:140-164
  A5  <image>/text splice                                       metamorph_arch.py:177-425
  A6  LLaMA decoder                                             (transformers LlamaModel)
  A7  lm_head + shifted CE                                      metamorph_llama.py:393-413
  A8  vision_head + normalise + cosine / soft-CE / mean-abs     metamorph_llama.py:420-462
  A9  loss combine incl. the NaN / "loss doubles" quirks        metamorph_llama.py:461-474

Tensors that require grad in `sd` receive gradients through ordinary torch autograd, so the
same function is the backward oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import ref_ops as ops
from .ref_plan import splice_bookkeeping, IGNORE_INDEX, IMAGE_START_ID


@dataclass
class OracleConfig:
    # LLM
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    vocab_size: int = 128258
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: dict | None = None                   # LlamaConfig.rope_scaling: {"rope_type": "llama3" | "linear", ...} (ref_ops.rope_inv_freq)
    max_position_embeddings: int = 8192
    head_dim_explicit: int | None = None               # LlamaConfig.head_dim when the checkpoint states it (q_proj is then [Hq * d, h], d != h / Hq)
    tie_word_embeddings: bool = False                  # lm_head.weight IS model.embed_tokens.weight (LLaMA-3.2 1B / 3B)
    # vision tower (SigLIP geometry; reference hard-codes SO400M/14-384, siglip_encoder.py:113)
    v_hidden: int = 1152
    v_layers: int = 27
    v_heads: int = 16
    v_intermediate: int = 4304
    v_patch: int = 14
    v_image: int = 384
    v_ln_eps: float = 1e-6
    # connector / heads
    num_image_tokens: int = 256
    mm_projector_type: str = "mlp2x_gelu"
    vision_head_type: str = "mlp"
    normalize_vision: bool = True
    apply_softmax: bool = False
    use_vision_ar: bool = True
    vision_coef: float = 1.0
    tokenizer_model_max_length: int | None = 4096
    tokenizer_padding_side: str = "right"
    image_start_id: int = IMAGE_START_ID
    image_token_reduction: str = "interpolation"       # | "mlpmixer" | "concat_interpolation" (siglip_encoder.py:151-204)
    pretraining_tp: int = 1                            # > 1: lm_head applied in vocabulary slices (metamorph_llama.py:393-396)

    @property
    def head_dim(self):
        return self.head_dim_explicit or self.hidden_size // self.num_attention_heads

    @property
    def rope_kw(self):
        return dict(rope_scaling=self.rope_scaling, max_position_embeddings=self.max_position_embeddings)


# ------------------------------------------------------------------ SigLIP tower (A3)

def siglip_hidden(sd, cfg: OracleConfig, images: torch.Tensor, prefix="model.vision_tower.vision_tower."):
    """images [N,3,H,W] -> last encoder-layer output [N,P,hv] (no post_layernorm, no head)."""
    g = lambda k: sd[prefix + k]
    w = g("embeddings.patch_embedding.weight")
    dt = w.dtype
    x = torch.nn.functional.conv2d(images.to(dt), w, g("embeddings.patch_embedding.bias"),
                                   stride=cfg.v_patch)
    N, C, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2) + g("embeddings.position_embedding.weight")[None]
    H, d = cfg.v_heads, cfg.v_hidden // cfg.v_heads
    for j in range(cfg.v_layers):
        p = f"encoder.layers.{j}."
        h = ops.layernorm(x, g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), cfg.v_ln_eps)
        q = ops.linear(h, g(p + "self_attn.q_proj.weight"), g(p + "self_attn.q_proj.bias"))
        k = ops.linear(h, g(p + "self_attn.k_proj.weight"), g(p + "self_attn.k_proj.bias"))
        v = ops.linear(h, g(p + "self_attn.v_proj.weight"), g(p + "self_attn.v_proj.bias"))
        sh = lambda t: t.view(N, -1, H, d).transpose(1, 2)
        a = ops.attention(sh(q), sh(k), sh(v), None, causal=False)
        a = a.transpose(1, 2).reshape(N, -1, H * d)
        x = x + ops.linear(a, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        h = ops.layernorm(x, g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"), cfg.v_ln_eps)
        h = ops.gelu_tanh(ops.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")))
        x = x + ops.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    return x


def reduce_features(sd, cfg: OracleConfig, f: torch.Tensor):
    """hidden_states[-1] of the tower [N, P, hv] -> [N, T, hv]: token reduction, L2 normalisation, softmax (siglip_encoder.py:151-211)."""
    if f.shape[1] != cfg.num_image_tokens and cfg.image_token_reduction == "mlpmixer":
        # token_mixer = Linear(P, T) over the patch axis, then channel_mixer = Linear(hv, hv)  (siglip_encoder.py:164-168)
        p = "model.vision_tower."
        f = ops.linear(f.transpose(1, 2), sd[p + "token_mixer.0.weight"], sd[p + "token_mixer.0.bias"]).transpose(1, 2)
        f = ops.linear(f, sd[p + "channel_mixer.0.weight"], sd[p + "channel_mixer.0.bias"])
    elif f.shape[1] != cfg.num_image_tokens and cfg.image_token_reduction == "concat_interpolation":
        # bilinear to 4 T tokens, then every 2 x 2 block of the grid concatenated along the channels (siglip_encoder.py:169-199)
        N, _, C = f.shape
        s = int(math.isqrt(cfg.num_image_tokens))
        g = ops.bilinear_reduce(f, 4 * cfg.num_image_tokens).view(N, 2 * s, 2 * s, C)
        g = g.view(N, s, 2, s, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, s * s, 4 * C)
        f = g
    else:
        f = ops.bilinear_reduce(f, cfg.num_image_tokens)
    if cfg.normalize_vision:
        f = ops.l2_normalize(f)
    if cfg.apply_softmax:
        f = torch.softmax(f / 0.07, dim=-1)
    return f


def vision_features(sd, cfg: OracleConfig, images: torch.Tensor, train_vision: bool = False):
    """SiglipVisionTower.forward: tower -> cast to images.dtype -> reduce -> normalise, under
    torch.set_grad_enabled(not freeze_vision) (reference siglip_encoder.py:138-139)."""
    with torch.set_grad_enabled(train_vision):
        return reduce_features(sd, cfg, siglip_hidden(sd, cfg, images).to(images.dtype))


# ------------------------------------------------------------------ projector / heads

def _mlp(sd, prefix, x, n_linear):
    """nn.Sequential(Linear, GELU, Linear, ...) with keys prefix{0,2,4}.{weight,bias}."""
    for i in range(n_linear):
        x = ops.linear(x, sd[f"{prefix}{2 * i}.weight"], sd[f"{prefix}{2 * i}.bias"])
        if i != n_linear - 1:
            x = ops.gelu_erf(x)
    return x


def mm_projector(sd, cfg: OracleConfig, feat):
    t = cfg.mm_projector_type
    if t == "linear":
        return ops.linear(feat, sd["model.mm_projector.weight"], sd["model.mm_projector.bias"])
    if t == "identity":
        return feat
    if t == "mlpsoftmax":                                    # Linear -> Softmax(dim=-1) -> Linear (multimodal_projector/builder.py:45-50)
        z = ops.linear(feat, sd["model.mm_projector.0.weight"], sd["model.mm_projector.0.bias"])
        return ops.linear(torch.softmax(z, dim=-1), sd["model.mm_projector.2.weight"], sd["model.mm_projector.2.bias"])
    if t.startswith("mlp") and t.endswith("x_gelu"):
        return _mlp(sd, "model.mm_projector.", feat, int(t[3:-6]))
    raise ValueError(f"Unknown projector type: {t}")


def vision_head(sd, cfg: OracleConfig, x):
    t = cfg.vision_head_type
    if t == "mlp":
        return _mlp(sd, "vision_head.", x, 2)
    if t == "mlp2x_gelu":
        return _mlp(sd, "vision_head.", x, 3)
    return ops.linear(x, sd["vision_head.weight"], sd["vision_head.bias"])   # "linear" / default


# ------------------------------------------------------------------ LLaMA decoder (A6)

def llama_layer(sd, cfg: OracleConfig, i: int, x, key_valid, cos, sin):
    """One HF LlamaDecoderLayer (index i): RMSNorm -> q/k/v -> RoPE -> causal SDPA (+ key padding) -> o_proj -> + residual ->
    RMSNorm -> SwiGLU -> + residual.  x [B, L, h]; cos / sin [B, L, d] from ref_ops.rope_tables."""
    B, L, h = x.shape
    Hq, Hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    p = f"model.layers.{i}."
    n = ops.rmsnorm(x, sd[p + "input_layernorm.weight"], cfg.rms_norm_eps)
    q = ops.linear(n, sd[p + "self_attn.q_proj.weight"]).view(B, L, Hq, d).transpose(1, 2)
    k = ops.linear(n, sd[p + "self_attn.k_proj.weight"]).view(B, L, Hkv, d).transpose(1, 2)
    v = ops.linear(n, sd[p + "self_attn.v_proj.weight"]).view(B, L, Hkv, d).transpose(1, 2)
    q, k = ops.rope_apply(q, cos, sin), ops.rope_apply(k, cos, sin)
    a = ops.attention(q, k, v, key_valid, causal=True)
    a = a.transpose(1, 2).reshape(B, L, Hq * d)
    x = x + ops.linear(a, sd[p + "self_attn.o_proj.weight"])
    n = ops.rmsnorm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
    g = ops.linear(n, sd[p + "mlp.gate_proj.weight"])
    u = ops.linear(n, sd[p + "mlp.up_proj.weight"])
    return x + ops.linear(ops.swiglu(g, u), sd[p + "mlp.down_proj.weight"])


def llama_decoder(sd, cfg: OracleConfig, x, key_valid, position_ids=None):
    B, L, h = x.shape
    if position_ids is None:
        position_ids = torch.arange(L)[None].expand(B, L)
    cos, sin = ops.rope_tables(position_ids, cfg.head_dim, cfg.rope_theta, x.dtype, **cfg.rope_kw)
    for i in range(cfg.num_hidden_layers):
        x = llama_layer(sd, cfg, i, x, key_valid, cos, sin)
    return ops.rmsnorm(x, sd["model.norm.weight"], cfg.rms_norm_eps)


# ------------------------------------------------------------------ splice (A5)

def splice(sd, cfg: OracleConfig, input_ids, labels, attention_mask, proj_feat, target_feat):
    """Returns inputs_embeds [B,L,h], labels [B,L] | None, attention_mask bool [B,L],
    image_positions [B,L] int64, target_features [Na,T,hv], position_ids [B,L]."""
    ids = input_ids.tolist()
    lab = labels.tolist() if labels is not None else None
    msk = attention_mask.bool().tolist() if attention_mask is not None else None
    N, T, _ = proj_feat.shape
    plan = splice_bookkeeping(ids, lab, msk, N, T, cfg.tokenizer_model_max_length,
                              cfg.tokenizer_padding_side, cfg.image_start_id)
    emb = sd["model.embed_tokens.weight"]
    B, L = len(plan["src"]), len(plan["src"][0])
    h = emb.shape[1]
    # row sources as the bookkeeping lists them: None = padding (zeros), (img, i, t) = row t of image i, int = token id.
    # Assembled with two gathers (one autograd node each) instead of one select per row: the backward of a per-row `emb[s]` allocates
    # a full [V, h] zero tensor PER ROW, minutes at V = 128258 -- same values, same gradients.
    tok_pos, tok_ids, img_pos, img_rows = [], [], [], []
    for b in range(B):
        for l, s in enumerate(plan["src"][b]):
            if s is None:
                continue
            if isinstance(s, tuple):
                img_pos.append(b * L + l)
                img_rows.append(s[1] * T + s[2])
            else:
                tok_pos.append(b * L + l)
                tok_ids.append(s)
    x = torch.zeros(B * L, h, dtype=proj_feat.dtype)
    if tok_pos:
        x = x.index_put((torch.tensor(tok_pos),), emb[torch.tensor(tok_ids)].to(proj_feat.dtype))
    if img_pos:
        x = x.index_put((torch.tensor(img_pos),), proj_feat.reshape(N * T, h)[torch.tensor(img_rows)])
    x = x.view(B, L, h)
    out_labels = torch.tensor(plan["labels"], dtype=torch.long) if plan["labels"] is not None else None
    return (x, out_labels, torch.tensor(plan["attention_mask"], dtype=torch.bool),
            torch.tensor(plan["image_positions"], dtype=torch.long),
            target_feat[plan["target_keep"]] if len(plan["target_keep"]) else target_feat[:0],
            torch.tensor(plan["position_ids"], dtype=torch.long))


# ------------------------------------------------------------------ full forward

def heads(sd, cfg: OracleConfig, hid, lab, img_pos, target, return_logits=True, ce_rows_only=False):
    """Everything after the decoder's final norm: lm_head + shifted CE (A7), vision_head + image-AR loss (A8), combine (A9).
    hid [B, L, h]; lab [B, L] | None; img_pos [B, L]; target [Na, T, hv].  Returns the loss entries of `forward`'s dict."""
    out = {}

    def lm_head(x):
        w = sd["lm_head.weight"]
        if cfg.pretraining_tp > 1:                              # :393-396: split along the vocabulary, one matmul per slice, concatenate
            slices = w.split(cfg.vocab_size // cfg.pretraining_tp, dim=0)
            return torch.cat([ops.linear(x, slices[i]) for i in range(cfg.pretraining_tp)], dim=-1)
        return ops.linear(x, w)

    if ce_rows_only and lab is not None and not return_logits:
        # full-size cases: logits only for the rows whose NEXT label is live -- the same mean NLL as below, without the
        # [B, L, V] fp32 tensor the reference materialises
        nxt = torch.full_like(lab, IGNORE_INDEX)
        nxt[:, :-1] = lab[:, 1:]
        keep = nxt != IGNORE_INDEX
        lg = lm_head(hid[keep]).float()
        ce = (torch.logsumexp(lg, -1) - lg.gather(1, nxt[keep][:, None])[:, 0]).sum() / keep.sum()
    else:
        logits = lm_head(hid).float()
        if return_logits:
            out["logits"] = logits
        if lab is None:
            out["loss"] = None
            return out
        ce = ops.shifted_cross_entropy(logits, lab, IGNORE_INDEX)
    if img_pos is None:                                         # no images given: the image-AR block is skipped altogether (:333, :420)
        out["loss"] = ce
        return out
    # rows of hidden[:, :-1] whose NEXT position is an answer-image row (metamorph_llama.py:386-390,425-432)
    sel = img_pos[:, 1:].bool()
    pred_in = hid[:, :-1][sel]                                  # [R,h], row-major (b,t) order
    pred = vision_head(sd, cfg, pred_in)
    if cfg.normalize_vision:
        pred = ops.l2_normalize(pred)
    if cfg.apply_softmax:
        pred = torch.softmax(pred / 0.07, dim=-1)
    tgt = target.reshape(-1, target.shape[-1])
    if cfg.apply_softmax:
        l_img = ops.soft_ce_loss(tgt, pred)
    elif cfg.normalize_vision:
        if tgt.shape != pred.shape:
            l_img = ce                                         # the try/except at :451-455 (row count or -- concat_interpolation -- width)
        else:
            l_img = ops.cosine_loss(tgt, pred)
    else:
        l_img = ops.mean_abs_loss(tgt, pred)
    out["loss_language"] = float(ce.detach())
    out["loss_image_ar"] = float(l_img.detach())
    loss = ce
    if cfg.use_vision_ar and float(l_img.detach()) != 0:
        loss = ce + cfg.vision_coef * l_img
    out["loss"] = loss
    out["pred"] = pred
    return out


def forward(sd, cfg: OracleConfig, input_ids, attention_mask, labels, images, return_logits=True, train_vision=False,
            ce_rows_only=False, image_embeds=None):
    if images is None and image_embeds is None:
        # prepare_inputs_labels_for_multimodal returns its inputs untouched (metamorph_arch.py:184-191): plain LLaMA forward over the padded
        # ids with the caller's mask, no image bookkeeping
        x = sd["model.embed_tokens.weight"][input_ids]
        key_valid = attention_mask.bool() if attention_mask is not None else torch.ones_like(input_ids, dtype=torch.bool)
        hid = llama_decoder(sd, cfg, x, key_valid, None)
        out = {"hidden_states": hid, "labels": labels, "attention_mask": key_valid, "image_positions": None, "target_features": None,
               "inputs_embeds": x}
        out.update(heads(sd, cfg, hid, labels, None, None, return_logits, ce_rows_only))
        return out
    if image_embeds is not None:                               # `encode_imagesembed` (metamorph_arch.py:166-173): features given, no tower
        feat = image_embeds
    else:
        feat = vision_features(sd, cfg, images, train_vision)  # [N,T,hv]; no grad unless the tower trains (row N4)
    proj = mm_projector(sd, cfg, feat)                         # [N,T,h]
    target = feat.detach().clone()
    x, lab, key_valid, img_pos, target, _pid = splice(sd, cfg, input_ids, labels, attention_mask, proj, target)
    # position_ids stays None in the reference when the caller passes None (metamorph_arch.py:411-412)
    hid = llama_decoder(sd, cfg, x, key_valid, None)
    out = {"hidden_states": hid, "labels": lab, "attention_mask": key_valid,
           "image_positions": img_pos, "target_features": target, "inputs_embeds": x}
    out.update(heads(sd, cfg, hid, lab, img_pos, target, return_logits, ce_rows_only))
    return out


# ------------------------------------------------------------------ greedy decode (row N1)

def greedy_decode(sd, cfg: OracleConfig, input_ids, images=None, max_new_tokens=1024, start_image_token_id=IMAGE_START_ID,
                  end_image_token_id=128257, eos_token_id=(128001, 128009)):
    """The reference's `generate` + `greedy_decode` loop (metamorph_llama.py:665-717, 502-597) restated: the prefix is re-run every
    step (use_cache=False there); in image mode the last hidden row goes through vision_head -> normalise (-> softmax / 0.07) ->
    mm_projector and is fed back as the next input row (:363-377).  Returns (token ids, pred_z [n, hv], per-step last-row logits)."""
    with torch.no_grad():
        if images is not None:
            feat = vision_features(sd, cfg, images)
            proj = mm_projector(sd, cfg, feat)
            x, _, _, _, _, _ = splice(sd, cfg, input_ids, None, None, proj, feat)
        else:
            x = sd["model.embed_tokens.weight"][input_ids]
        in_image_mode = False
        generated, embeds, step_logits = [], [], []
        total_image_tokens = total_out = 0
        while True:
            B, L, _ = x.shape
            hid = llama_decoder(sd, cfg, x, torch.ones(B, L, dtype=torch.bool))
            image_embed = None
            if in_image_mode:
                pred_z = vision_head(sd, cfg, hid[:, -1, :])
                if cfg.normalize_vision:
                    pred_z = ops.l2_normalize(pred_z)
                if cfg.apply_softmax:
                    pred_z = torch.softmax(pred_z / 0.07, dim=-1)
                image_embed = pred_z
                hid = hid.clone()
                hid[:, -1, :] = mm_projector(sd, cfg, pred_z)
            logits = ops.linear(hid[:, -1, :], sd["lm_head.weight"]).float()
            step_logits.append(logits[0])
            next_token = int(torch.argmax(logits, dim=-1)[0])
            next_embed = hid[:, -1:, :]
            tok_embed = sd["model.embed_tokens.weight"][torch.tensor([[next_token]])].to(x.dtype)
            if (not in_image_mode) and next_token == start_image_token_id:
                in_image_mode = True
                generated.append(next_token)
                x = torch.cat((x, tok_embed), dim=1)
            elif in_image_mode and total_image_tokens < cfg.num_image_tokens:
                total_image_tokens += 1
                embeds.append(image_embed)
                x = torch.cat((x, next_embed), dim=1)
                if total_image_tokens == cfg.num_image_tokens:
                    in_image_mode = False
            elif next_token == end_image_token_id:
                in_image_mode = False
                total_image_tokens = 0
                generated.append(next_token)
                x = torch.cat((x, tok_embed), dim=1)
            else:
                x = torch.cat((x, tok_embed), dim=1)
                generated.append(next_token)
            total_out += 1
            if next_token in eos_token_id or total_out > max_new_tokens:
                break
        pz = torch.cat(embeds, dim=0) if embeds else torch.zeros(0)
        return generated, pz, step_logits


def decode_fixture_state_dict(g, cfg: OracleConfig, dtype=torch.float32):
    """Weights of a tests/golden/n1_decode_*.npz fixture: the seeded state dict with the sparse lm_head the fixture describes
    (oracle/gen_golden.py:gen_decode -- zero rows except `active` (8 x seeded) and the solved `row_tokens` rows)."""
    import numpy as np
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    W = torch.zeros_like(sd["lm_head.weight"])
    act = g["active"].tolist()
    W[act] = 8.0 * sd["lm_head.weight"][act]
    for tok, vec in zip(g["row_tokens"].tolist(), g["row_values"]):
        W[tok] = torch.from_numpy(np.asarray(vec))
    sd["lm_head.weight"] = W
    return {k: v.to(dtype) for k, v in sd.items()}


# ------------------------------------------------------------------ synthetic weights

def init_state_dict(cfg: OracleConfig, seed: int, dtype=torch.float32, std: float = 0.02,
                    with_vision=True, fast_big=False):
    """Deterministic N(0, std) weights (numpy PCG64 so the stream is platform independent);
    norm weights 1 + N(0, 0.1) so that their gradients are exercised.  fast_big=True draws tensors of more than 2^20 elements
    from torch's multi-threaded generator instead (full-width test cases: ~1.5 G values; the device model and the oracle
    are built from the same dict in the same process, so platform independence is not needed there)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    tg = torch.Generator().manual_seed(seed)
    sd = {}

    def rnd(*shape, s=std):
        n = 1
        for d in shape:
            n *= d
        if fast_big and n > (1 << 20):
            return (torch.randn(*shape, generator=tg, dtype=torch.float32) * s).to(dtype)
        return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * s).to(dtype)

    def norm_w(n):
        return (1.0 + rnd(n, s=0.1).float()).to(dtype)

    h, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    d, Hq, Hkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    sd["model.embed_tokens.weight"] = rnd(V, h)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = rnd(Hq * d, h)
        sd[p + "self_attn.k_proj.weight"] = rnd(Hkv * d, h)
        sd[p + "self_attn.v_proj.weight"] = rnd(Hkv * d, h)
        sd[p + "self_attn.o_proj.weight"] = rnd(h, Hq * d)
        sd[p + "mlp.gate_proj.weight"] = rnd(I, h)
        sd[p + "mlp.up_proj.weight"] = rnd(I, h)
        sd[p + "mlp.down_proj.weight"] = rnd(h, I)
        sd[p + "input_layernorm.weight"] = norm_w(h)
        sd[p + "post_attention_layernorm.weight"] = norm_w(h)
    sd["model.norm.weight"] = norm_w(h)
    hv = cfg.v_hidden                                     # tower feature width = vision_head output width (reference literal 1152)
    hv_mm = hv * (4 if cfg.image_token_reduction == "concat_interpolation" else 1)              # mm_hidden_size: projector input
    n_lin = {"linear": 1, "identity": 0, "mlpsoftmax": 2}.get(cfg.mm_projector_type)
    if n_lin is None:
        n_lin = int(cfg.mm_projector_type[3:-6])
    if cfg.mm_projector_type == "linear":
        sd["model.mm_projector.weight"] = rnd(h, hv_mm); sd["model.mm_projector.bias"] = rnd(h)
    else:
        for j in range(n_lin):
            sd[f"model.mm_projector.{2 * j}.weight"] = rnd(h, hv_mm if j == 0 else h)
            sd[f"model.mm_projector.{2 * j}.bias"] = rnd(h)
    sd["model.vision_proj.weight"] = rnd(h, 4096)       # dead Linear(4096, h), metamorph_arch.py:31
    sd["model.vision_proj.bias"] = rnd(h)
    sd["lm_head.weight"] = rnd(V, h)
    if cfg.tie_word_embeddings:                           # one tensor under both names (the draw above keeps the stream of the other keys)
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    if cfg.vision_head_type == "mlp":
        sd["vision_head.0.weight"] = rnd(h, h); sd["vision_head.0.bias"] = rnd(h)
        sd["vision_head.2.weight"] = rnd(hv, h); sd["vision_head.2.bias"] = rnd(hv)
    elif cfg.vision_head_type == "mlp2x_gelu":
        sd["vision_head.0.weight"] = rnd(h, h); sd["vision_head.0.bias"] = rnd(h)
        sd["vision_head.2.weight"] = rnd(h, h); sd["vision_head.2.bias"] = rnd(h)
        sd["vision_head.4.weight"] = rnd(hv, h); sd["vision_head.4.bias"] = rnd(hv)
    elif cfg.vision_head_type == "linear":
        sd["vision_head.weight"] = rnd(h, h); sd["vision_head.bias"] = rnd(h)
    else:
        sd["vision_head.weight"] = rnd(hv, h); sd["vision_head.bias"] = rnd(hv)
    if with_vision:
        vp = "model.vision_tower.vision_tower."
        P = (cfg.v_image // cfg.v_patch) ** 2
        if cfg.image_token_reduction == "mlpmixer":
            sd["model.vision_tower.token_mixer.0.weight"] = rnd(cfg.num_image_tokens, P, s=0.2)
            sd["model.vision_tower.token_mixer.0.bias"] = rnd(cfg.num_image_tokens, s=0.2)
            sd["model.vision_tower.channel_mixer.0.weight"] = rnd(hv, hv)
            sd["model.vision_tower.channel_mixer.0.bias"] = rnd(hv)
        sd[vp + "embeddings.patch_embedding.weight"] = rnd(hv, 3, cfg.v_patch, cfg.v_patch)
        sd[vp + "embeddings.patch_embedding.bias"] = rnd(hv)
        sd[vp + "embeddings.position_embedding.weight"] = rnd(P, hv)
        for j in range(cfg.v_layers):
            p = vp + f"encoder.layers.{j}."
            for ln in ("layer_norm1", "layer_norm2"):
                sd[p + ln + ".weight"] = norm_w(hv)
                sd[p + ln + ".bias"] = rnd(hv)
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[p + f"self_attn.{nm}.weight"] = rnd(hv, hv)
                sd[p + f"self_attn.{nm}.bias"] = rnd(hv)
            sd[p + "mlp.fc1.weight"] = rnd(cfg.v_intermediate, hv)
            sd[p + "mlp.fc1.bias"] = rnd(cfg.v_intermediate)
            sd[p + "mlp.fc2.weight"] = rnd(hv, cfg.v_intermediate)
            sd[p + "mlp.fc2.bias"] = rnd(hv)
    return sd
