#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE ITSELF (TEST INFRASTRUCTURE).

Runs only in the build container (needs /root/reference; the GPU box never has it).  It imports
the reference's own Python -- metamorph.mm_utils, metamorph.model.* -- plus the third-party
`transformers` the reference delegates its LLaMA/SigLIP arithmetic to, feeds them seeded inputs
and random weights, and records inputs/outputs as small .npz/.json fixtures.  No reference source
text is written anywhere; fixtures are data only.

    python oracle/gen_golden.py            # rewrites ALL of tests/golden/ (bit-identical to the committed files)
    python oracle/gen_golden.py e2e a5     # only these generators

Shims: import-only `wandb`/`decord` packages from oracle/_shims, and a tiny
SiglipVisionModel assigned to the delay-loaded tower (the checkpoint cannot be downloaded).
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import numpy as np
import torch

from oracle.fake_tokenizer import FakeTokenizer
from oracle.gen_inputs import a5_random_batches
from oracle.ref_model import OracleConfig, init_state_dict

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

torch.set_num_threads(8)


def save_npz(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().to(torch.float32).numpy() if v.is_floating_point() else v.detach().numpy()
        conv[k] = v
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **conv)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------- A1

A1_PROMPTS = [
    "hello world",
    "<image>",
    "<image> describe this picture",
    "describe <image>",
    "a <image> b <image> c",
    "<image><image>",
    "<image> <image> tail",
    "<|begin_of_text|> system prompt <image_start><image><image_end> what is this",
    "<|begin_of_text|><|start_header_id|> user <|end_header_id|> draw a cat <|eot_id|>"
    "<|start_header_id|> assistant <|end_header_id|> sure <image_start><image><image_end><|eot_id|>",
    "",
    "x <image>",
    "<image> y <image>",
]


def gen_a1():
    from metamorph.mm_utils import tokenizer_image_token
    cases = []
    for add_bos in (True, False):
        tok = FakeTokenizer(add_bos=add_bos)
        for p in A1_PROMPTS:
            for idx in (-200, -7):
                ids = tokenizer_image_token(p, tok, image_token_index=idx)
                cases.append(dict(prompt=p, add_bos=add_bos, image_token_index=idx, ids=ids))
    with open(os.path.join(OUT, "a1_tokenizer.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print(f"  wrote a1_tokenizer.json ({len(cases)} cases)")


# ----------------------------------------------------------------------------- reference model builder

def build_reference(cfg: OracleConfig, sd, dtype, **ctor):
    from transformers import SiglipVisionConfig, SiglipVisionModel
    from metamorph.model.language_model.metamorph_llama import MetaMorphLlamaForCausalLM, MetaMorphConfig

    hf = MetaMorphConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                         num_key_value_heads=cfg.num_key_value_heads, vocab_size=cfg.vocab_size,
                         max_position_embeddings=cfg.max_position_embeddings, rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps,
                         attention_bias=False, tie_word_embeddings=cfg.tie_word_embeddings,
                         **({"rope_scaling": dict(cfg.rope_scaling)} if cfg.rope_scaling else {}),
                         **({"head_dim": cfg.head_dim_explicit} if cfg.head_dim_explicit else {}))
    hf.mm_vision_tower = "siglip/CLIP-ViT-SO400M-14-384"
    hf.mm_projector_type = cfg.mm_projector_type
    hf.mm_hidden_size = cfg.v_hidden * (4 if cfg.image_token_reduction == "concat_interpolation" else 1)
    hf.num_image_tokens = cfg.num_image_tokens
    hf.image_token_reduction = cfg.image_token_reduction
    hf.freeze_vision = True
    hf.normalize_vision = cfg.normalize_vision
    hf.apply_softmax = cfg.apply_softmax
    hf.mm_vision_select_layer = -1
    hf.tokenizer_model_max_length = cfg.tokenizer_model_max_length
    hf.tokenizer_padding_side = cfg.tokenizer_padding_side
    hf.vision_head_type = cfg.vision_head_type
    model = MetaMorphLlamaForCausalLM(hf, use_vision_ar=cfg.use_vision_ar, vision_head=cfg.vision_head_type,
                                      vision_coef=cfg.vision_coef, normalize_vision=cfg.normalize_vision,
                                      apply_softmax=cfg.apply_softmax, vision_delay_load=True, **ctor)
    tower = model.get_model().vision_tower
    vcfg = SiglipVisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_intermediate,
                              num_hidden_layers=cfg.v_layers, num_attention_heads=cfg.v_heads,
                              image_size=cfg.v_image, patch_size=cfg.v_patch, layer_norm_eps=cfg.v_ln_eps,
                              hidden_act="gelu_pytorch_tanh")
    tower.vision_tower = SiglipVisionModel(vcfg)
    tower.is_loaded = True
    if cfg.image_token_reduction == "mlpmixer":              # what the tower's constructor builds once it is loaded (:100-107)
        P = (cfg.v_image // cfg.v_patch) ** 2
        tower.token_mixer = torch.nn.Sequential(torch.nn.Linear(P, cfg.num_image_tokens))
        tower.channel_mixer = torch.nn.Sequential(torch.nn.Linear(cfg.v_hidden, cfg.v_hidden))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "post_layernorm" not in k and ".head." not in k]
    assert not missing and not unexpected, (missing, unexpected)
    model.to(dtype)
    model.train()
    return model


def tiny_cfg(**kw):
    base = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=1, vocab_size=128258, v_layers=2, v_intermediate=144, v_image=56,
                num_image_tokens=4, tokenizer_model_max_length=64)
    base.update(kw)
    return OracleConfig(**base)


# ----------------------------------------------------------------------------- A5 cases

ST, IM, EN = 128256, -200, 128257


def a5_cases():
    """(name, ids rows, label rows, T, max_len, padding_side).  Rows are ragged; padded below."""
    A = 128000
    c = []
    # prompt-side image then text answer; answer-side image; text-only with dummy image
    c.append(("mixed_right", [
        [A, A, 11, 12, ST, IM, EN, 13, 14, 15, 16],
        [A, A, 21, 22, 23, ST, IM, EN, 128009],
        [A, A, 31, 32, 33, 34],
    ], [
        [-100] * 7 + [13, 14, 15, 16],
        [-100] * 5 + [ST, IM, EN, 128009],
        [-100, -100, -100, 32, 33, 34],
    ], 4, 64, "right"))
    c.append(("mixed_left", c[0][1], c[0][2], 4, 64, "left"))
    # two images in one sample: prompt image + answer image; second sample two prompt images
    c.append(("two_images", [
        [A, A, 5, ST, IM, EN, 6, 7, ST, IM, EN, 8],
        [A, ST, IM, EN, 9, ST, IM, EN, 10, 11],
    ], [
        [-100] * 7 + [7, ST, IM, EN, 8],
        [-100] * 8 + [10, 11],
    ], 4, 64, "right"))
    # overflow: second image does not fit -> dropped with all later text (need_to_stop)
    c.append(("overflow_drop", [
        [A, A, 5, ST, IM, EN, 6, 7, ST, IM, EN, 8, 9],
        [A, A, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, ST, IM, EN, 51],
    ], [
        [-100] * 7 + [7, ST, IM, EN, 8, 9],
        [-100] * 2 + [41, 42, 43, 44, 45, 46, 47, 48, 49, 50, ST, IM, EN, 51],
    ], 16, 24, "right"))
    # text longer than max_len -> truncation after the loop
    c.append(("truncate_text", [
        [A, A] + list(range(100, 130)),
        [A, A, 7, ST, IM, EN, 8],
    ], [
        [-100, -100] + list(range(100, 130)),
        [-100] * 6 + [8],
    ], 4, 20, "right"))
    # answer image is the last thing that fits exactly
    c.append(("exact_fit", [
        [A, A, 1, 2, ST, IM, EN],
    ], [
        [-100, -100, -100, 2, ST, IM, EN],
    ], 4, 9, "right"))
    # image first in the sequence after BOS only, answer-image decided by label just before it
    c.append(("label_rule", [
        [A, ST, IM, EN, 3],
        [A, 77, IM, EN, 3],
        [A, ST, IM, 4],
    ], [
        [-100, ST, IM, EN, 3],
        [-100, 77, IM, EN, 3],
        [-100, -100, IM, 4],
    ], 4, 64, "right"))
    return c


def pad_rows(rows, pad):
    T = max(len(r) for r in rows)
    return [r + [pad] * (T - len(r)) for r in rows]


def gen_a5():
    cfg = tiny_cfg(hidden_size=32, intermediate_size=64, num_attention_heads=2, num_key_value_heads=1)
    sd = init_state_dict(cfg, seed=11)
    rng = np.random.default_rng(5)
    pad_id = 128001

    def record(name, ids, labs, T, max_len, side, use_labels=True, use_mask=True, pos=None, pad_rows_with=pad_id):
        cfg.num_image_tokens = T
        cfg.tokenizer_model_max_length = max_len
        cfg.tokenizer_padding_side = side
        model = build_reference(cfg, sd, torch.float32)
        if max_len is None:                                   # a config without the attribute: no truncation, no overflow rule (:271)
            del model.config.tokenizer_model_max_length
        n_img = sum(max(1, sum(1 for t in r if t == IM)) for r in ids)
        images = torch.from_numpy(rng.standard_normal((n_img, 3, 56, 56), dtype=np.float32))
        ids_t = torch.tensor(pad_rows(ids, pad_rows_with))
        lab_t = torch.tensor(pad_rows(labs, -100))
        msk_t = ids_t.ne(pad_id)
        pos_t = None if pos is None else torch.tensor(pos)
        with torch.no_grad():
            proj, feat = model.encode_images(images)
            try:
                out = model.prepare_inputs_labels_for_multimodal(ids_t, pos_t, msk_t if use_mask else None, None, lab_t if use_labels else None, images)
            except TypeError as e:                            # known answer: no tokenizer_model_max_length + an image -> `int > None` (:324)
                save_npz(f"a5_{name}.npz", input_ids=ids_t, labels=lab_t, attention_mask=msk_t, rows_per_image=np.int64(T), max_length=np.int64(-1),
                         left=np.int64(side == "left"), num_images=np.int64(n_img), error=np.array(type(e).__name__))
                return
        _, pos_ids, att, _, emb, new_lab, img_pos, tgt = out
        # derive, from the reference's own output, where every spliced row came from
        W = model.get_model().embed_tokens.weight.detach()
        flat_proj = proj.reshape(-1, proj.shape[-1])
        B, L, _ = emb.shape
        src = np.full((B, L), -1, dtype=np.int64)
        for b in range(B):
            toks = [t for t in ids_t[b].tolist() if t >= 0]
            for l in range(L):
                row = emb[b, l]
                if torch.count_nonzero(row) == 0:             # padding rows (zero embeddings)
                    assert att is None or not att[b, l]
                    continue
                hit = (flat_proj == row).all(dim=1).nonzero()
                if len(hit):
                    src[b, l] = -2 - int(hit[0, 0])
                    continue
                cand = [t for t in set(toks) if torch.equal(W[t], row)]
                assert len(cand) == 1, (name, b, l, cand)
                src[b, l] = cand[0]
        keep = []
        for r in range(tgt.shape[0]):
            hit = [i for i in range(feat.shape[0]) if torch.equal(feat[i], tgt[r])]
            assert len(hit) == 1
            keep.append(hit[0])
        if use_labels and use_mask and pos is None and max_len is not None:     # the original seven cases: same keys, same order as before
            rec = dict(input_ids=ids_t, labels=lab_t, attention_mask=msk_t,
                       rows_per_image=np.int64(T), max_length=np.int64(max_len), left=np.int64(side == "left"),
                       num_images=np.int64(n_img), out_src=src, out_labels=new_lab, out_attention_mask=att,
                       out_image_positions=img_pos, out_position_ids_is_none=np.int64(pos_ids is None),
                       out_target_keep=np.array(keep, dtype=np.int64))
        else:
            rec = dict(input_ids=ids_t, labels=lab_t, attention_mask=msk_t,
                       rows_per_image=np.int64(T), max_length=np.int64(max_len if max_len is not None else -1), left=np.int64(side == "left"),
                       num_images=np.int64(n_img), out_src=src, out_image_positions=img_pos, out_position_ids_is_none=np.int64(pos_ids is None),
                       out_target_keep=np.array(keep, dtype=np.int64))
            rec.update(labels_given=np.int64(use_labels), mask_given=np.int64(use_mask), out_labels_is_none=np.int64(new_lab is None),
                       out_mask_is_none=np.int64(att is None), out_shape=np.array(emb.shape[:2]))
            if new_lab is not None:
                rec["out_labels"] = new_lab
            if att is not None:
                rec["out_attention_mask"] = att
                rec["out_attention_mask_dtype"] = np.array(str(att.dtype))
            if pos is not None:
                rec["position_ids"] = pos_t
                rec["out_position_ids"] = pos_ids
        save_npz(f"a5_{name}.npz", **rec)

    for name, ids, labs, T, max_len, side in a5_cases():
        record(name, ids, labs, T, max_len, side)
    # the wrapper's None handling (metamorph_arch.py:245-256, 400-412), recorded after the original cases (same image stream, their files unchanged)
    base = a5_cases()[0]
    nopad = ([[128000, 128000, 11, ST, IM, EN, 13, 14], [128000, 21, 22, 23, ST, IM, EN, 128009]],
             [[-100] * 6 + [13, 14], [-100] * 4 + [ST, IM, EN, 128009]])
    record("wrap_nolabels", base[1], base[2], 4, 64, "right", use_labels=False)                   # inference: labels None -> None back
    record("wrap_nomask", nopad[0], nopad[1], 4, 64, "right", use_mask=False)                     # attention_mask None -> None back
    record("wrap_nomask_padded", base[1], base[2], 4, 64, "right", use_mask=False)                # ... and pad ids are then EMBEDDED like any token
    pos = [list(range(7, 7 + len(pad_rows(base[1], pad_id)[0])))] * 3
    record("wrap_position_ids", base[1], base[2], 4, 64, "right", pos=pos)                        # given position_ids are REPLACED by arange per sample
    record("wrap_position_ids_left", base[1], base[2], 4, 64, "left", pos=pos)
    record("wrap_no_max_length", a5_cases()[3][1], a5_cases()[3][2], 16, None, "right")           # no tokenizer_model_max_length + images: the reference raises
    record("wrap_no_max_length_text", [[128000, 5, 6, 7], [128000, 8]], [[-100, 5, 6, 7], [-100, 8]], 4, None, "right")   # ... text-only rows never reach that line


def gen_a5rand():
    """120 random batches through the REFERENCE's `prepare_inputs_labels_for_multimodal` (metamorph_arch.py:177-425): labels, attention mask,
    image positions, kept targets and the origin of every spliced row, padded to common shapes in one .npz."""
    cfg = tiny_cfg(hidden_size=32, intermediate_size=64, num_attention_heads=2, num_key_value_heads=1)
    sd = init_state_dict(cfg, seed=11)
    rng = np.random.default_rng(6)
    pad_id = 128001
    cases = a5_random_batches()
    N = len(cases)
    Lmax = max(min(max(len(r) + 16 * r.count(-200) for r in rows), mx) for rows, _, _, mx, _ in cases)
    out_lab = np.full((N, 4, Lmax), -9, dtype=np.int64)
    out_msk = np.full((N, 4, Lmax), -9, dtype=np.int64)
    out_pos = np.full((N, 4, Lmax), -9, dtype=np.int64)
    out_src = np.full((N, 4, Lmax), -9, dtype=np.int64)
    shape = np.zeros((N, 2), dtype=np.int64)
    keep = np.full((N, 16), -9, dtype=np.int64)
    models = {}
    for ci, (rows, labs, T, max_len, side) in enumerate(cases):
        key = (T, max_len, side)
        if key not in models:
            cfg.num_image_tokens, cfg.tokenizer_model_max_length, cfg.tokenizer_padding_side = T, max_len, side
            models[key] = build_reference(cfg, sd, torch.float32)
        model = models[key]
        n_img = sum(max(1, r.count(-200)) for r in rows)
        images = torch.from_numpy(rng.standard_normal((n_img, 3, 56, 56), dtype=np.float32))
        ids_t, lab_t = torch.tensor(pad_rows(rows, pad_id)), torch.tensor(pad_rows(labs, -100))
        msk_t = ids_t.ne(pad_id)
        with torch.no_grad():
            proj, feat = model.encode_images(images)
            _, _, att, _, emb, new_lab, img_pos, tgt = model.prepare_inputs_labels_for_multimodal(ids_t, None, msk_t, None, lab_t, images)
        B, L, _ = emb.shape
        W = model.get_model().embed_tokens.weight.detach()
        flat_proj = proj.reshape(-1, proj.shape[-1])
        src = np.full((B, L), -1, dtype=np.int64)
        for b in range(B):
            toks = set(t for t in rows[b] if t >= 0)
            for l in range(L):
                if not att[b, l]:
                    continue
                row = emb[b, l]
                hit = (flat_proj == row).all(dim=1).nonzero()
                if len(hit):
                    src[b, l] = -2 - int(hit[0, 0])
                    continue
                cand = [t for t in toks if torch.equal(W[t], row)]
                assert len(cand) == 1, (ci, b, l, cand)
                src[b, l] = cand[0]
        kp = []
        for r in range(tgt.shape[0]):
            hit = [i for i in range(feat.shape[0]) if torch.equal(feat[i], tgt[r])]
            assert len(hit) == 1
            kp.append(hit[0])
        shape[ci] = (B, L)
        out_lab[ci, :B, :L], out_msk[ci, :B, :L], out_pos[ci, :B, :L], out_src[ci, :B, :L] = new_lab.numpy(), att.numpy(), img_pos.numpy(), src
        keep[ci, :len(kp)] = kp
    save_npz("a5rand_reference.npz", n_cases=np.int64(N), seed=np.int64(0), out_shape=shape, out_labels=out_lab, out_attention_mask=out_msk,
             out_image_positions=out_pos, out_src=out_src, out_target_keep=keep)
    print(f"    {N} random batches; output lengths {int(shape[:, 1].min())}..{int(shape[:, 1].max())}; "
          f"{int((keep >= 0).sum())} kept targets; {sum(1 for c in cases if c[4] == 'left')} left-padded")


# ----------------------------------------------------------------------------- A3 tower

def gen_a3():
    for T in (4, 16):
        cfg = tiny_cfg(num_image_tokens=T)
        sd = init_state_dict(cfg, seed=21)
        rng = np.random.default_rng(22)
        images = torch.from_numpy(rng.standard_normal((2, 3, 56, 56), dtype=np.float32))
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            model = build_reference(cfg, sd, dt)
            with torch.no_grad():
                tower = model.get_model().vision_tower
                raw = tower.vision_tower(images.to(dt), output_hidden_states=True).hidden_states[-1]
                feat = tower(images.to(dt))
                proj, tgt = model.encode_images(images.to(dt))
            save_npz(f"a3_tower_T{T}_{tag}.npz", images=images, seed=np.int64(21), raw_hidden=raw[:, :, ::8],
                     features=feat, projected=proj[:, :, ::4], target=tgt[:, :, ::16])


def gen_a3sel():
    """Two more branches of `SiglipVisionTower.forward` (siglip_encoder.py:129-150): `mm_vision_select_layer = -2` (hidden_states[-2] = the
    output of the encoder layer before the last) and `num_image_tokens = -1` (the tower returns ZEROS of the un-reduced shape)."""
    cfg = tiny_cfg(num_image_tokens=4, v_layers=3)
    sd = init_state_dict(cfg, seed=23)
    rng = np.random.default_rng(24)
    images = torch.from_numpy(rng.standard_normal((2, 3, 56, 56), dtype=np.float32))
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        model = build_reference(cfg, sd, dt)
        tower = model.get_model().vision_tower
        with torch.no_grad():
            tower.select_layer = -2
            hs = tower.vision_tower(images.to(dt), output_hidden_states=True).hidden_states
            feat = tower(images.to(dt))
            tower.select_layer = -1
            tower.image_token_len = -1
            zeros = tower(images.to(dt))
        assert not torch.equal(hs[-2], hs[-1])
        save_npz(f"a3sel_tower_{tag}.npz", images=images, seed=np.int64(23), raw_hidden_m2=hs[-2][:, :, ::8], features_m2=feat,
                 tokens_minus1=zeros[:, :, ::64], tokens_minus1_shape=np.array(zeros.shape), tokens_minus1_absmax=zeros.abs().max())


# ----------------------------------------------------------------------------- per-op (v)

def gen_ops():
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, LlamaRotaryEmbedding, apply_rotary_pos_emb, LlamaConfig
    import torch.nn.functional as F
    rng = np.random.default_rng(31)
    r = lambda *s: torch.from_numpy(rng.standard_normal(s, dtype=np.float32))
    out = {}
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        x = r(3, 7, 256).to(dt)
        w = (1 + 0.1 * r(256)).to(dt)
        n = LlamaRMSNorm(256, eps=1e-5).to(dt)
        n.weight.data.copy_(w)
        out[f"rms_x_{tag}"], out[f"rms_w_{tag}"], out[f"rms_y_{tag}"] = x, w, n(x)
        # rope
        cfg = LlamaConfig(hidden_size=256, num_attention_heads=2, rope_theta=500000.0, max_position_embeddings=8192)
        rot = LlamaRotaryEmbedding(cfg)
        q, k = r(2, 2, 9, 128).to(dt), r(2, 1, 9, 128).to(dt)
        pos = torch.arange(9)[None].expand(2, 9)
        cos, sin = rot(q, pos)
        qe, ke = apply_rotary_pos_emb(q, k, cos, sin)
        out[f"rope_q_{tag}"], out[f"rope_k_{tag}"], out[f"rope_qe_{tag}"], out[f"rope_ke_{tag}"] = q, k, qe, ke
        out[f"rope_cos_{tag}"], out[f"rope_sin_{tag}"] = cos, sin
        # causal + key padding SDPA, GQA 2:1 (additive min-dtype mask exactly as HF builds it)
        L = 9
        v = r(2, 1, L, 128).to(dt)
        valid = torch.tensor([[True] * 9, [True] * 6 + [False] * 3])
        allow = torch.ones(L, L, dtype=torch.bool).tril()[None, None] & valid[:, None, None, :]
        kk = k.repeat_interleave(2, dim=1)
        vv = v.repeat_interleave(2, dim=1)
        att = F.scaled_dot_product_attention(qe, ke.repeat_interleave(2, dim=1), vv, attn_mask=allow, scale=128 ** -0.5)
        out[f"att_v_{tag}"], out[f"att_valid_{tag}"], out[f"att_o_{tag}"] = v, valid, att
        # swiglu, gelus, layernorm
        g, u = r(5, 64).to(dt), r(5, 64).to(dt)
        out[f"swi_g_{tag}"], out[f"swi_u_{tag}"], out[f"swi_y_{tag}"] = g, u, F.silu(g) * u
        out[f"gelu_erf_{tag}"] = torch.nn.GELU()(g)
        out[f"gelu_tanh_{tag}"] = F.gelu(g, approximate="tanh")
        lw, lb = (1 + 0.1 * r(64)).to(dt), (0.1 * r(64)).to(dt)
        out[f"ln_w_{tag}"], out[f"ln_b_{tag}"], out[f"ln_y_{tag}"] = lw, lb, F.layer_norm(g, (64,), lw, lb, 1e-6)
        # interpolation 27x27 -> 16x16 / 8x8 and 4x4 -> 2x2 + normalise (siglip_encoder.py:160-163,208)
        for side_in, side_out in ((27, 16), (27, 8), (4, 2)):
            f = r(2, side_in * side_in, 24).to(dt)
            y = f.view(2, side_in, side_in, 24).permute(0, 3, 1, 2).contiguous()
            y = F.interpolate(y.to(torch.float32), size=(side_out, side_out), mode="bilinear", align_corners=False).to(dt)
            y = y.permute(0, 2, 3, 1).contiguous().flatten(1, 2)
            out[f"interp_{side_in}_{side_out}_x_{tag}"] = f
            out[f"interp_{side_in}_{side_out}_y_{tag}"] = y
            out[f"interp_{side_in}_{side_out}_yn_{tag}"] = F.normalize(y, p=2, dim=-1)
        # CE (shifted, ignore -100, mean) and cosine
        lg = r(2, 6, 50).to(dt).float()
        lb_ = torch.tensor([[-100, 3, 4, -100, 7, 49], [-100, -100, 1, 2, -100, -100]])
        ce = torch.nn.CrossEntropyLoss()(lg[:, :-1].reshape(-1, 50), lb_[:, 1:].reshape(-1))
        out[f"ce_logits_{tag}"], out[f"ce_labels_{tag}"], out[f"ce_loss_{tag}"] = lg, lb_, ce
        a, b = F.normalize(r(6, 32), dim=-1).to(dt), F.normalize(r(6, 32), dim=-1).to(dt)
        out[f"cos_t_{tag}"], out[f"cos_p_{tag}"] = a, b
        out[f"cos_loss_{tag}"] = -F.cosine_similarity(a, b, dim=-1).mean()
    # AdamW: 3 steps of torch.optim.AdamW
    p = torch.nn.Parameter(r(257))
    opt = torch.optim.AdamW([p], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    out["adam_p0"] = p.detach().clone()
    gs = []
    for s in range(3):
        gr = r(257)
        gs.append(gr)
        p.grad = gr.clone()
        opt.step()
    out["adam_grads"] = torch.stack(gs)
    out["adam_p3"] = p.detach().clone()
    save_npz("ops.npz", **out)


# ----------------------------------------------------------------------------- end to end (vi, vii)

def e2e_batch(kind):
    A = 128000
    if kind == "mixed":
        ids = [[A, A, 11, 12, ST, IM, EN, 13, 14, 15, 16, 128009],
               [A, A, 21, 22, 23, ST, IM, EN, 128009],
               [A, A, 31, 32, 33, 34, 35]]
        lab = [[-100] * 7 + [13, 14, 15, 16, 128009],
               [-100] * 5 + [ST, IM, EN, 128009],
               [-100, -100, -100, 32, 33, 34, 35]]
    elif kind == "understanding_only":       # no answer-side image -> loss_image_ar = NaN (A9)
        ids = [[A, A, 11, ST, IM, EN, 13, 14],
               [A, A, 31, 32, 33, 34]]
        lab = [[-100] * 6 + [13, 14],
               [-100, -100, -100, 32, 33, 34]]
    elif kind == "generation_only":
        ids = [[A, A, 21, 22, ST, IM, EN, 128009],
               [A, A, 23, ST, IM, EN, 128009, 128001]]
        lab = [[-100] * 4 + [ST, IM, EN, 128009],
               [-100] * 3 + [ST, IM, EN, 128009, -100]]
    elif kind == "multi_frame":
        # BASELINE configs[2] in miniature: several prompt-side frames per sample (interleaved "video" frames), one sample that
        # also carries an answer-side image after three prompt frames, one with two frames and trailing text
        ids = [[A, A, 11, ST, IM, EN, ST, IM, EN, ST, IM, EN, 12, 13, 14, 128009],
               [A, A, ST, IM, EN, 21, ST, IM, EN, ST, IM, EN, 22, ST, IM, EN, 128009],
               [A, A, 31, ST, IM, EN, 32, ST, IM, EN, 33, 34, 35]]
        lab = [[-100] * 12 + [12, 13, 14, 128009],
               [-100] * 12 + [22, ST, IM, EN, 128009],
               [-100] * 10 + [33, 34, 35]]
    else:
        raise KeyError(kind)
    return ids, lab


def grad_summary(t):
    f = t.detach().float().flatten()
    n = min(256, f.numel())
    idx = (torch.arange(n, dtype=torch.long) * (f.numel() - 1)) // max(n - 1, 1)
    return torch.cat([f.norm()[None], f[idx]])


ROPE31 = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
ROPE31_TINY = dict(ROPE31, original_max_position_embeddings=32)
E2E_CASES = [
    # (kind, T, use_vision_ar, head variant): "cos" = normalize_vision (every shipped recipe); "l1" = the constructor default
    # (mean-abs `mse_loss_fn`); "softce" = apply_softmax on normalised features; "softce_raw" = apply_softmax alone
    ("mixed", 4, True, "cos"), ("mixed", 16, True, "cos"), ("understanding_only", 4, True, "cos"),
    ("understanding_only", 4, False, "cos"), ("generation_only", 4, True, "cos"),
    ("multi_frame", 4, True, "cos"),
    ("mixed", 4, True, "l1"), ("generation_only", 4, True, "l1"),
    ("mixed", 4, True, "softce"), ("generation_only", 4, True, "softce_raw"),
    ("mixed", 4, True, "cos", "left"),               # tokenizer_padding_side = "left" (metamorph_arch.py:362-386)
    ("mixed", 4, True, "cos", "right", {"mm_projector_type": "mlpsoftmax"}),      # Linear -> Softmax -> Linear connector
    ("mixed", 1, True, "cos", "right", {"image_token_reduction": "concat_interpolation"}),   # 16 patches -> 4 -> one 4 x 1152-wide token
    ("understanding_only", 4, False, "cos", "right", {"image_token_reduction": "mlpmixer"}),
    # round 3: the remaining connector / head types the reference builds (builder.py:38-62, metamorph_llama.py:246-269)
    ("mixed", 4, True, "cos", "right", {"mm_projector_type": "linear"}),
    ("mixed", 4, True, "cos", "right", {"mm_projector_type": "mlp3x_gelu"}),
    ("mixed", 4, True, "cos", "right", {"vision_head_type": "mlp2x_gelu"}),
    ("mixed", 4, True, "cos", "right", {"vision_head_type": "linear"}),     # Linear(h, h): width != 1152 -> the try/except fallback L_img = CE (:451-455)
    ("generation_only", 4, True, "cos", "right", {"vision_head_type": "None"}),   # the constructor default: one Linear(h, 1152)
    ("mixed", 4, True, "cos", "right", {"vision_coef": 0.25}),                   # loss = CE + 0.25 * L_img (:470-474)
    ("mixed", 4, False, "cos"),                      # use_vision_ar = False WITH answer images: L_img is computed and logged, not added (:470)
    # depth: 8 decoder layers (4 heads over 1 KV head, d = 64) and 4 tower layers on the multi-frame batch -- the oracle's layer stacking
    # (and its layer-streamed form, oracle/ref_stream.py) pinned to the reference beyond the 2-layer fixtures
    ("multi_frame", 4, True, "cos", "right", {"num_hidden_layers": 8, "v_layers": 4, "num_attention_heads": 4}),
    # the 'identity' connector (builder.py:60-61) needs an LLM as wide as the tower: h = 1152 = 9 heads x 128 over 3 KV heads
    ("mixed", 4, True, "cos", "right", {"mm_projector_type": "identity", "hidden_size": 1152, "num_attention_heads": 9, "num_key_value_heads": 3}),
    # round 6: the config fields a real base checkpoint carries.  LLaMA-3.1's RoPE (rope_type "llama3": the reference README's own base
    # model, README.md:178,187; inherited through MetaMorphConfig(LlamaConfig), metamorph_llama.py:129-133) with the pre-training context
    # shrunk to 32 positions so that all three wavelength bands (kept < 8, blended 8..32, stretched > 32) act within these 12..21-row
    # samples (d = 128: bands 0-1 kept, 2-7 blended, 8-63 stretched); the real 3.1 constants at L = 4096 are gen_rope31's long case
    ("mixed", 4, True, "cos", "right", {"rope_scaling": ROPE31_TINY, "max_position_embeddings": 256}, "_rope-llama3"),
    ("multi_frame", 4, True, "cos", "left", {"rope_scaling": ROPE31_TINY, "max_position_embeddings": 256}, "_left_rope-llama3"),
    ("mixed", 4, True, "cos", "right", {"rope_scaling": {"rope_type": "linear", "factor": 4.0}}, "_rope-linear"),
    ("mixed", 4, True, "cos", "right", {"tie_word_embeddings": True}, "_tied"),              # lm_head.weight IS embed_tokens.weight (LLaMA-3.2 1B / 3B)
    # explicit head_dim != hidden_size / heads: 2 heads x 64 over a 256-wide model (q_proj [128, 256], o_proj [256, 128])
    ("mixed", 4, True, "cos", "right", {"head_dim_explicit": 64}, "_headdim64"),
]
HEAD_VARIANTS = {"cos": (True, False), "l1": (False, False), "softce": (True, True), "softce_raw": (False, True)}


def gen_e2e(first=0):
    shared = np.random.default_rng(41)               # the first five cases draw their images from ONE stream, in this order
    for i, case in enumerate(E2E_CASES):
        if i < first:                                # (`first` > 5 only: the shared stream is consumed in order)
            assert first > 5
            continue
        kind, T, use_ar, variant = case[:4]
        side = case[4] if len(case) > 4 else "right"
        extra = case[5] if len(case) > 5 else {}
        rng = shared if i < 5 else np.random.default_rng(4100 + i)
        nv, sm = HEAD_VARIANTS[variant]
        cfg = tiny_cfg(num_image_tokens=T, use_vision_ar=use_ar, normalize_vision=nv, apply_softmax=sm, tokenizer_padding_side=side, **extra)
        sd = init_state_dict(cfg, seed=43)
        ids, lab = e2e_batch(kind)
        pad_id = 128001
        ids_t = torch.tensor(pad_rows(ids, pad_id))
        lab_t = torch.tensor(pad_rows(lab, -100))
        msk_t = ids_t.ne(pad_id)
        n_img = sum(max(1, sum(1 for t in r if t == IM)) for r in ids)
        images = torch.from_numpy(rng.standard_normal((n_img, 3, 56, 56), dtype=np.float32))
        suffix = ("" if variant == "cos" else "_" + variant) + ("" if side == "right" else "_" + side) + "".join(
            "_" + ("head-" if k == "vision_head_type" else "coef" if k == "vision_coef" else "") + str(v) for k, v in extra.items())
        structural = {k: v for k, v in extra.items() if k in ("num_hidden_layers", "v_layers", "num_attention_heads", "num_key_value_heads", "hidden_size",
                                                              "rope_scaling", "max_position_embeddings", "tie_word_embeddings", "head_dim_explicit")}
        if len(case) > 6:
            suffix = case[6]
        elif "num_hidden_layers" in structural:
            suffix = f"_deep{extra['num_hidden_layers']}"
        elif structural:
            suffix = "_" + str(extra["mm_projector_type"])
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            model = build_reference(cfg, sd, dt)
            for n, p in model.named_parameters():
                p.requires_grad_("vision_tower" not in n)
            out = model(input_ids=ids_t, attention_mask=msk_t, labels=lab_t, images=images.to(dt))
            rec = dict(input_ids=ids_t, labels=lab_t, attention_mask=msk_t, images=images,
                       seed=np.int64(43), rows_per_image=np.int64(T), use_vision_ar=np.int64(use_ar),
                       normalize_vision=np.int64(nv), apply_softmax=np.int64(sm), left=np.int64(side == "left"),
                       mm_projector_type=np.array(cfg.mm_projector_type), image_token_reduction=np.array(cfg.image_token_reduction),
                       **({"vision_head_type": np.array(cfg.vision_head_type)} if "vision_head_type" in extra else {}),
                       **({"vision_coef": np.float64(cfg.vision_coef)} if "vision_coef" in extra else {}),
                       **({"cfg_json": np.array(json.dumps(structural))} if structural else {}),
                       target_features_shape=np.array(model.prepare_inputs_labels_for_multimodal(ids_t, None, msk_t, None, lab_t, images.to(dt))[7].shape),
                       loss=out.loss.detach().float(), loss_language=np.float64(model.loss_language),
                       loss_image_ar=np.float64(model.loss_image_ar),
                       logits_sub=out.logits[:, :, ::997], hidden=out.hidden_states)
            if torch.isfinite(out.loss):
                out.loss.backward()
                for n, p in model.named_parameters():
                    if p.grad is not None and "vision_proj" not in n:
                        rec["grad::" + n] = grad_summary(p.grad)
            save_npz(f"e2e_{kind}_T{T}_ar{int(use_ar)}{suffix}_{tag}.npz", **rec)


# ----------------------------------------------------------------------------- N2 (batch producer)

N2_SOURCES = {
    "text_1round": [[{"from": "human", "value": "what is the capital of france"}, {"from": "gpt", "value": "paris of course"}]],
    "text_batch2": [[{"from": "human", "value": "count to three"}, {"from": "gpt", "value": "one two three"}],
                    [{"from": "human", "value": "say hi"}, {"from": "gpt", "value": "hi there my friend how are you today"}]],
    "image_prompt": [[{"from": "human", "value": "<image>\nwhat is shown here"}, {"from": "gpt", "value": "a small red bird"}]],
    "image_answer_2rounds": [[{"from": "human", "value": "hello"}, {"from": "gpt", "value": "hello there"},
                              {"from": "human", "value": "draw a cat on a mat"}, {"from": "gpt", "value": "here it is <image>"}]],
    "two_images_mixed": [[{"from": "human", "value": "<image> make it blue"}, {"from": "gpt", "value": "<image> done"}]],
    "starts_with_gpt": [[{"from": "gpt", "value": "ignored greeting"}, {"from": "human", "value": "tell me a joke"},
                         {"from": "gpt", "value": "knock knock"}]],
    "open_answer": [[{"from": "human", "value": "<image> describe"}, {"from": "gpt", "value": ""}]],
    "three_rounds": [[{"from": "human", "value": "a"}, {"from": "gpt", "value": "b c"}, {"from": "human", "value": "d e f"},
                      {"from": "gpt", "value": "g"}, {"from": "human", "value": "h"}, {"from": "gpt", "value": "i j k l"}]],
}


def gen_n2():
    """preprocess_multimodal + preprocess_llama3 + DataCollatorForSupervisedDataset of the reference (train.py:309-332,
    501-597, 1251-1284) on the fake tokenizer; fixtures are the conversations and the integer outputs."""
    import copy
    import transformers
    import transformers.pytorch_utils
    import transformers.trainer as tr
    if not hasattr(tr, "ALL_LAYERNORM_LAYERS"):
        tr.ALL_LAYERNORM_LAYERS = transformers.pytorch_utils.ALL_LAYERNORM_LAYERS
    import metamorph.train.train as T
    from types import SimpleNamespace
    cases = []
    for name, sources in N2_SOURCES.items():
        for add_bos in (True, False):
            for use_se in (True, False):
                for max_len in (4096, 12):
                    has_image = any("<image>" in m["value"] for src in sources for m in src)
                    if not has_image and use_se:
                        continue
                    tok = FakeTokenizer(add_bos=add_bos, model_max_length=max_len)
                    src = copy.deepcopy(sources)
                    src = T.preprocess_multimodal(src, SimpleNamespace(is_multimodal=True, mm_use_im_start_end=use_se))
                    try:
                        out = T.preprocess_llama3(copy.deepcopy(src), tok, has_image=has_image)
                    except Exception as e:            # e.g. stacking ragged image prompts: the reference raises
                        cases.append(dict(name=name, sources=sources, add_bos=add_bos, mm_use_im_start_end=use_se,
                                          model_max_length=max_len, has_image=has_image, raises=type(e).__name__))
                        continue
                    cases.append(dict(name=name, sources=sources, add_bos=add_bos, mm_use_im_start_end=use_se,
                                      model_max_length=max_len, has_image=has_image,
                                      input_ids=out["input_ids"].tolist(), labels=out["labels"].tolist()))
    # collator: ragged instances, truncation, images as lists of tensors
    coll = []
    g = torch.Generator().manual_seed(0)
    for max_len in (4096, 6):
        tok = FakeTokenizer(model_max_length=max_len)
        inst = []
        for n, nimg in ((5, 1), (9, 2), (3, 0)):
            ids = torch.randint(3, 1000, (n,), generator=g)
            lab = ids.clone(); lab[: n // 2] = -100
            d = dict(input_ids=ids, labels=lab, image=[torch.full((3, 2, 2), float(10 * len(inst) + k)) for k in range(nimg)])
            inst.append(d)
        out = T.DataCollatorForSupervisedDataset(tokenizer=tok)(inst)
        coll.append(dict(model_max_length=max_len, pad_token_id=tok.pad_token_id,
                         instances=[dict(input_ids=d["input_ids"].tolist(), labels=d["labels"].tolist(),
                                         image=[im.tolist() for im in d["image"]]) for d in inst],
                         out=dict(input_ids=out["input_ids"].tolist(), labels=out["labels"].tolist(),
                                  attention_mask=out["attention_mask"].tolist(), images=out["images"].tolist())))
    with open(os.path.join(OUT, "n2_batch_producer.json"), "w") as f:
        json.dump(dict(preprocess=cases, collate=coll), f, indent=0)
    print(f"  wrote n2_batch_producer.json ({len(cases)} preprocess cases, {len(coll)} collate cases)")


def gen_n2rand():
    """40 seeded random conversations (oracle/gen_inputs.n2_random_sources) through the reference's preprocess_multimodal + preprocess_llama3
    (train.py:309-332, 501-597) on the fake tokenizer, five tokenizer / template settings each: ids and labels (inputs are regenerated from
    the seed by the test, only outputs are stored)."""
    import copy
    from types import SimpleNamespace
    from oracle.gen_inputs import n2_random_sources
    T, _ = _import_train()
    cases = []
    for si, sources in enumerate(n2_random_sources()):
        has_image = any("<image>" in m["value"] for src in sources for m in src)
        for add_bos, use_se, max_len in ((True, False, 4096), (False, False, 4096), (True, True, 4096), (False, True, 4096), (True, False, 24)):
            tok = FakeTokenizer(add_bos=add_bos, model_max_length=max_len)
            rec = dict(source=si, add_bos=add_bos, mm_use_im_start_end=use_se, model_max_length=max_len, has_image=has_image)
            try:
                src = T.preprocess_multimodal(copy.deepcopy(sources), SimpleNamespace(is_multimodal=True, mm_use_im_start_end=use_se))
                out = T.preprocess_llama3(copy.deepcopy(src), tok, has_image=has_image)
                rec.update(input_ids=out["input_ids"].tolist(), labels=out["labels"].tolist())
            except Exception as e:
                rec["raises"] = type(e).__name__
            cases.append(rec)
    with open(os.path.join(OUT, "n2_random.json"), "w") as f:
        json.dump({"seed": 7, "n_sources": 40, "cases": cases}, f)
    n_masked = sum(1 for c in cases if "labels" in c and all(x == -100 for x in c["labels"][0]))
    print(f"    {len(cases)} cases, {sum('raises' in c for c in cases)} raise, {sum(c['has_image'] for c in cases)} with images, {n_masked} fully masked (mismatch rule)")


def gen_images():
    """process_images / expand2square of the reference (mm_utils.py:151-188) driving transformers' own SigLIP image processor
    (the class the reference obtains from the hub, built here from the so400m-patch14-384 preprocessor constants at a small
    output size so that the fixture stays small): uint8 inputs -> float pixel tensors, 'pad' and default aspect modes."""
    from types import SimpleNamespace
    from PIL import Image
    from transformers import SiglipImageProcessor as HFProc
    from metamorph.mm_utils import process_images, expand2square
    size = 24
    hf = HFProc(do_resize=True, size={"height": size, "width": size}, resample=3, do_rescale=True, rescale_factor=1 / 255,
                do_normalize=True, image_mean=[0.5] * 3, image_std=[0.5] * 3)
    hf.crop_size = {"height": size, "width": size}
    rs = np.random.RandomState(7)
    raw = {"wide": (rs.rand(10, 31, 3) * 255).astype(np.uint8), "tall": (rs.rand(33, 12, 3) * 255).astype(np.uint8),
           "square": (rs.rand(24, 24, 3) * 255).astype(np.uint8), "gray": (rs.rand(17, 9) * 255).astype(np.uint8)}
    pil = {k: Image.fromarray(v) for k, v in raw.items()}
    out = {}
    for k, v in raw.items():
        out["in_" + k] = v
    names = sorted(raw)
    rgb = [pil[k].convert("RGB") for k in names]
    out["pad"] = process_images(rgb, hf, SimpleNamespace(image_aspect_ratio="pad")).numpy()
    out["plain"] = process_images([pil[k] for k in names], hf, SimpleNamespace(image_aspect_ratio=None)).numpy()
    bg = tuple(int(x * 255) for x in hf.image_mean)
    for k in ("wide", "tall"):
        out["square_of_" + k] = np.asarray(expand2square(pil[k], bg))
    out["names"] = np.array(names)
    out["size"] = np.int64(size)
    save_npz("n2_process_images.npz", **out)


def gen_conv():
    """Prompts of the reference's conversation templates (conversation.py) for a few message lists."""
    import metamorph.conversation as C
    dialogs = {
        "open_turn": [(0, "What is in the picture?"), (1, None)],
        "image_first": [(0, "<image>\nDescribe."), (1, "A cat."), (0, "And now?"), (1, None)],
        "closed": [(0, "Hi"), (1, "Hello there.")],
        "tuple_message": [(0, ("Look <image> here", "IMG", "Default")), (1, None)],
        "empty": [],
    }
    cases = []
    for tname in ("llama3", "chatml_direct", "mistral_direct"):
        for dname, turns in dialogs.items():
            conv = C.conv_templates[tname].copy()
            for r, m in turns:
                conv.append_message(conv.roles[r], m)
            cases.append(dict(template=tname, dialog=dname, turns=[[r, list(m) if isinstance(m, tuple) else m] for r, m in turns],
                              prompt=conv.get_prompt(), roles=list(conv.roles), sep=conv.sep, system=conv.system,
                              version=conv.version))
    with open(os.path.join(OUT, "n2_conversation.json"), "w") as f:
        json.dump(dict(cases=cases, default=C.default_conversation.version), f, indent=0)
    print(f"  wrote n2_conversation.json ({len(cases)} prompts)")


# ----------------------------------------------------------------------------- N1: the reference's own greedy_decode loop

DECODE_ACTIVE = [ST, EN, 128009, 11, 12, 13, 14, 15, 40, 41, 42, 43, 44, 45, 46, 47]


def decode_lm_head(sd, rows):
    """lm_head of the decode fixtures: zero everywhere except a handful of ACTIVE token rows (8 x the seeded rows), so that
    argmax decisions have margins far above bf16 noise instead of being near-ties among 128k random logits; `rows` overrides
    individual rows (the <image_start> / <image_end> / <eot> rows aligned with recorded hidden rows, see gen_decode)."""
    W = torch.zeros_like(sd["lm_head.weight"])
    W[DECODE_ACTIVE] = 8.0 * sd["lm_head.weight"][DECODE_ACTIVE]
    for tok, vec in rows.items():
        W[tok] = vec
    return W


DECODE_SEEDS = {"image_prompt_rope31": 64}   # (seed 63 leaves the last decision a 1.7-logit margin: below the 2.0 bar)


def gen_decode(only=None):
    """Runs the reference's `generate` -> `greedy_decode` (metamorph_llama.py:502-597, 665-717; image-mode feedback :363-377)
    unchanged on a tiny random model and records the emitted token ids, the per-step top-2 logit margins and the predicted visual
    embeddings (`pred_z`).  The weights are the seeded ones except for lm_head (see decode_lm_head): the rows of the tokens the
    loop is meant to emit are solved (least norm) so that each scores 12 on the hidden row the reference itself produces at its
    step and 0 at every other step -- the loop then walks token mode -> <image_start> -> image mode (4 continuous tokens fed back
    through vision_head / mm_projector) -> <image_end> -> two text tokens -> <|eot_id|> with decision margins far above bf16 noise.
    (The hidden row of step k depends only on the tokens emitted before k, never on lm_head, so the construction converges by
    extending the correct prefix one step per pass.)"""
    A = 128000
    # round 6: the same walk under LLaMA-3.1's RoPE (ROPE31_TINY: all three wavelength bands act within these ~20 positions) -- prompt pass and
    # every cached step read the scaled tables
    rope31 = {"rope_scaling": ROPE31_TINY, "max_position_embeddings": 256}
    cases = {"text": ([[A, A, 11, 12, 13, 14, 15]], 0, {}), "image_prompt": ([[A, A, 11, ST, IM, EN, 12, 13]], 1, {}),
             "image_prompt_rope31": ([[A, A, 11, ST, IM, EN, 12, 13]], 1, rope31)}
    plan = [(0, ST), (5, EN), (6, 41), (7, 42), (8, 128009)]     # (loop iteration, token it must emit); iterations 1-4 = image mode
    image_steps = [1, 2, 3, 4]
    for ci, (name, (ids, n_img, extra)) in enumerate(cases.items()):
        if only is not None and name not in only:
            continue
        ids_t = torch.tensor(ids)
        seed = DECODE_SEEDS.get(name, 61 + ci)
        cfg = tiny_cfg(num_image_tokens=4, **extra)
        sd = init_state_dict(cfg, seed=seed)
        images = None
        if n_img:
            images = torch.from_numpy(np.random.default_rng(seed).standard_normal((n_img, 3, 56, 56), dtype=np.float32))

        def run(rows, dt, max_new=12):
            sd2 = dict(sd)
            sd2["lm_head.weight"] = decode_lm_head(sd, rows)
            model = build_reference(cfg, sd2, dt)
            model.eval()
            lm_in, logits = [], []
            model.lm_head.register_forward_pre_hook(lambda m, a: lm_in.append(a[0][0, -1].detach().float().clone()))
            model.lm_head.register_forward_hook(lambda m, a, o: logits.append(o[0, -1].detach().float().clone()))
            with torch.no_grad():
                out, emb = model.generate(inputs=ids_t, images=None if images is None else images.to(dt), output_image=True,
                                          max_new_tokens=max_new)
            return out[0].tolist(), emb.float(), lm_in, logits

        def solve(hid):
            """Rows for the planned tokens from the hidden rows known so far: 12 at the own step, 0 at the other known steps
            (eos: -0.5 inside image mode, where the loop only tests for eos)."""
            steps = sorted(hid)
            H = torch.stack([hid[t] for t in steps]).double()                     # [n_steps, h]
            pinv = torch.linalg.pinv(H)                                            # [h, n_steps]
            rows = {}
            for t, tok in plan:
                if t not in hid:
                    continue
                y = torch.tensor([12.0 if u == t else (-0.5 if (tok == 128009 and u in image_steps) else 0.0) for u in steps],
                                 dtype=torch.float64)
                rows[tok] = (pinv @ y).float()
            return rows

        hid, rows = {}, {}
        for _ in range(len(plan) + 1):
            toks, emb, lm_in, logits = run(rows, torch.float32)
            argm = [int(l.argmax()) for l in logits]
            good = 0                                                               # iterations whose emitted prefix is as planned
            want = dict(plan)
            for t in range(len(lm_in)):
                hid[t] = lm_in[t]                                                  # valid: everything before t was as planned
                if t in want and argm[t] != want[t]:
                    break
                good = t + 1
            if good >= plan[-1][0] + 1:
                break
            hid = {t: v for t, v in hid.items() if t <= good}
            rows = solve(hid)
        toks, emb, lm_in, logits = run(rows, torch.float32)
        assert toks == [tok for _, tok in plan] and emb.shape[0] == 4 and len(logits) == plan[-1][0] + 1, (name, toks, emb.shape)
        top2 = [l.topk(2) for l in logits]
        margins = torch.stack([t.values[0] - t.values[1] for t in top2])
        decision = [t for t, _ in plan]
        assert float(margins[decision].min()) > 2.0, margins
        for t in image_steps:          # eos must lose clearly inside image mode (128001 has an all-zero row: it ties with id 0, which
            assert float(logits[t][128009]) < float(logits[t].max()) - 0.3        # argmax returns first, so it can never be emitted)
        toks16, emb16, _, logits16 = run(rows, torch.bfloat16)
        assert toks16 == toks
        # `max_new_tokens` cuts the loop after max_new + 1 iterations, image-mode iterations included (:587-590): 2 -> <image_start> + two
        # continuous tokens, 6 -> through <image_end> and the first text token
        cut = {}
        for mn in (2, 6):
            t_m, e_m, _, l_m = run(rows, torch.float32, max_new=mn)
            cut[f"tokens_max{mn}"] = np.array(t_m, dtype=np.int64)
            cut[f"n_pred_z_max{mn}"] = np.int64(e_m.shape[0])
            cut[f"iterations_max{mn}"] = np.int64(len(l_m))
            assert e_m.shape[0] == 0 or torch.equal(e_m, emb[:e_m.shape[0]])
        print(f"    {name}: cut runs { {k: v.tolist() for k, v in cut.items()} }")
        save_npz(f"n1_decode_{name}.npz", seed=np.int64(seed), input_ids=ids_t,
                 images=images if images is not None else torch.zeros(0), active=np.array(DECODE_ACTIVE, dtype=np.int64),
                 row_tokens=np.array(list(rows.keys()), dtype=np.int64), row_values=torch.stack(list(rows.values())),
                 tokens=np.array(toks, dtype=np.int64), tokens_bf16=np.array(toks16, dtype=np.int64), margins=margins,
                 decision_steps=np.array(decision, dtype=np.int64),
                 step_argmax=np.array([int(t.indices[0]) for t in top2], dtype=np.int64),
                 active_logits=torch.stack([l[DECODE_ACTIVE] for l in logits]),
                 active_logits_bf16=torch.stack([l[DECODE_ACTIVE] for l in logits16]),
                 pred_z=emb, pred_z_bf16=emb16, max_new_tokens=np.int64(12), **cut,
                 **({"cfg_json": np.array(json.dumps(extra))} if extra else {}))
        print(f"    {name}: seed {seed} tokens {toks} min decision margin {float(margins[decision].min()):.3f} "
              f"pred_z bf16-vs-f32 rel {float((emb16 - emb).norm() / emb.norm()):.3e}")


def gen_names():
    """A10: the `--vision_tower` name grammar (reference siglip_encoder.py:34-59), outputs and errors."""
    from metamorph.model.multimodal_encoder.siglip_encoder import extract_res_interp
    cases = ["siglip/CLIP-ViT-SO400M-14-384", "timm/ViT-SO400M-14-SigLIP-384-res512", "siglip/CLIP-ViT-SO400M-14-res384-interp144",
             "timm/ViT-SO400M-14-SigLIP-interp64-res224", "siglip/CLIP-ViT-SO400M-14", "siglip/CLIP-ViT-SO400M-14-384-interp256",
             "siglip/CLIP-ViT-SO400M-14-res12-res34", "openai/clip-vit-large", "siglip/CLIP-ViT-SO400M-14-resnet",
             "timm/ViT-SO400M-14-SigLIP-interp"]
    out = []
    for n in cases:
        try:
            r = list(extract_res_interp(n))
        except ValueError:
            r = "ValueError"
        out.append([n, r])
    with open(os.path.join(OUT, "a10_tower_names.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"    {len(out)} names")


# ----------------------------------------------------------------------------- round 3: N3 layouts, image_embeds, pretraining_tp, mean-abs rows

def _import_train():
    import transformers
    import transformers.pytorch_utils
    import transformers.trainer as tr
    if not hasattr(tr, "ALL_LAYERNORM_LAYERS"):
        tr.ALL_LAYERNORM_LAYERS = transformers.pytorch_utils.ALL_LAYERNORM_LAYERS
    import metamorph.train.train as T
    import metamorph.train.metamorph_trainer as MT
    return T, MT


def gen_n3():
    """Row N3 pinned to the reference: runs the reference's own `safe_save_model_for_hf_trainer` (train.py:186-222) and
    `MetaMorphTrainer._save_checkpoint` / `_save` (metamorph_trainer.py:273-298) on a tiny model through a stub trainer object and
    records WHAT THEY WRITE: relative file paths, the key list / dtypes / shapes inside every `.bin`, and a checksum of each tensor;
    then `initialize_vision_modules(pretrain_mm_mlp_adapter=...)` (metamorph_arch.py:91-96) loading the adapter back into a fresh
    model (which keys it consumes).  `deepspeed` is an import-only shim (maybe_zero_3 takes its plain-tensor branch)."""
    import tempfile
    from types import SimpleNamespace
    T, MT = _import_train()
    cfg = tiny_cfg(num_image_tokens=4)
    sd = init_state_dict(cfg, seed=47)
    out = {"seed": 47, "cases": []}

    def listing(root):
        files = {}
        for d, _, fs in os.walk(root):
            for f in fs:
                rel = os.path.relpath(os.path.join(d, f), root)
                entry = {"bytes_nonzero": os.path.getsize(os.path.join(d, f)) > 0}
                if f.endswith(".bin"):
                    blob = torch.load(os.path.join(d, f), map_location="cpu", weights_only=True)
                    entry["keys"] = list(blob.keys())
                    entry["dtypes"] = [str(v.dtype) for v in blob.values()]
                    entry["shapes"] = [list(v.shape) for v in blob.values()]
                    entry["sums"] = [float(v.double().sum()) for v in blob.values()]
                files[rel] = entry
        return dict(sorted(files.items()))

    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        for use_se in (False, True):
            model = build_reference(cfg, sd, dt)
            # stage-1 freeze policy (train.py:1515-1519): only the projector (+ embeddings with start/end tokens) trains
            for n, p in model.named_parameters():
                p.requires_grad_("mm_projector" in n or (use_se and "embed_tokens" in n))
            for folder in ("final_model", "checkpoint-7"):
                with tempfile.TemporaryDirectory() as tmp:
                    outdir = os.path.join(tmp, "run", folder)
                    args = SimpleNamespace(tune_mm_mlp_adapter=True, use_im_start_end=use_se, local_rank=0, should_save=True, output_dir=os.path.join(tmp, "run"))
                    trainer = SimpleNamespace(args=args, model=model, deepspeed=None)
                    T.safe_save_model_for_hf_trainer(trainer, outdir)
                    out["cases"].append({"fn": "safe_save_model_for_hf_trainer", "dtype": tag, "use_im_start_end": use_se,
                                         "output_dir": f"run/{folder}", "files": listing(tmp)})
            with tempfile.TemporaryDirectory() as tmp:
                run = os.path.join(tmp, "run")
                args = SimpleNamespace(tune_mm_mlp_adapter=True, use_im_start_end=use_se, local_rank=-1, should_save=True, output_dir=run)
                stub = SimpleNamespace(args=args, model=model, state=SimpleNamespace(global_step=12), _get_output_dir=lambda trial=None: run)
                MT.MetaMorphTrainer._save_checkpoint(stub, model, None)
                MT.MetaMorphTrainer._save(stub, os.path.join(run, "ignored"))      # adapter runs: `_save` writes nothing (:294-298)
                out["cases"].append({"fn": "MetaMorphTrainer._save_checkpoint", "dtype": tag, "use_im_start_end": use_se,
                                     "global_step": 12, "files": listing(tmp)})
        # full model (stage 2): the non-deepspeed branch hands a CPU state dict to trainer._save (train.py:213-222)
        model = build_reference(cfg, sd, dt)
        captured = {}
        args = SimpleNamespace(tune_mm_mlp_adapter=False, should_save=True, local_rank=0)
        trainer = SimpleNamespace(args=args, model=model, deepspeed=None,
                                  _save=lambda output_dir, state_dict=None: captured.update(dir=output_dir, sd=state_dict))
        T.safe_save_model_for_hf_trainer(trainer, "some/dir")
        keys = list(captured["sd"].keys())
        out["cases"].append({"fn": "safe_save_model_for_hf_trainer(full)", "dtype": tag, "n_keys": len(keys),
                             "keys_without_tower": [k for k in keys if "vision_tower" not in k],
                             "all_cpu": all(v.device.type == "cpu" for v in captured["sd"].values()),
                             "dtypes": sorted({str(v.dtype) for v in captured["sd"].values()})})
    # re-load: pretrain_mm_mlp_adapter consumed by initialize_vision_modules (metamorph_arch.py:91-96)
    import tempfile as _tf
    with _tf.TemporaryDirectory() as tmp:
        donor = build_reference(cfg, init_state_dict(cfg, seed=48), torch.float32)
        blob = T.get_mm_adapter_state_maybe_zero_3(donor.named_parameters(), ["mm_projector", "embed_tokens"])
        path = os.path.join(tmp, "mm_projector.bin")
        torch.save(blob, path)
        fresh = build_reference(cfg, sd, torch.float32)
        margs = SimpleNamespace(vision_tower="siglip/CLIP-ViT-SO400M-14-384", mm_vision_select_layer=-1, mm_vision_select_feature="patch",
                                pretrain_mm_mlp_adapter=path, mm_projector_type="mlp2x_gelu", mm_patch_merge_type="flat",
                                image_token_reduction="interpolation", num_image_tokens=4, freeze_vision=True, normalize_vision=True,
                                apply_softmax=False, vision_coef=1.0)
        before = {k: v.detach().clone() for k, v in fresh.state_dict().items()}
        fresh.get_model().vision_tower.load_model = lambda *a, **k: None      # the hub download inside (:63) cannot run offline; the tower is built
        try:
            fresh.get_model().initialize_vision_modules(margs, fsdp=None)
            err = None
        except Exception as e:                                      # recorded, not hidden
            err = f"{type(e).__name__}: {e}"
        after = fresh.state_dict()
        changed = sorted(k for k in before if k in after and not torch.equal(before[k], after[k]))
        from_blob = sorted(k for k in changed if k in blob and torch.equal(after[k], blob[k].to(after[k].dtype)))
        out["reload"] = {"adapter_keys": list(blob.keys()), "error": err, "changed_keys": changed, "changed_to_adapter_values": from_blob}
    with open(os.path.join(OUT, "n3_checkpoint_layouts.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"  wrote n3_checkpoint_layouts.json ({len(out['cases'])} cases; reload changed {out['reload']['changed_keys']} error={out['reload']['error']})")


def gen_r3():
    """(i) `mse_loss_fn` (metamorph_llama.py:211-219) with equal and unequal row counts; (ii) forward(image_embeds=...) /
    `encode_imagesembed` (metamorph_arch.py:166-173, metamorph_llama.py:603-660); (iii) `pretraining_tp = 2` lm_head slicing
    (metamorph_llama.py:393-396): loss / logits / hidden / gradient summaries like the e2e fixtures."""
    from metamorph.model.language_model.metamorph_llama import mse_loss_fn
    import torch.nn.functional as F
    rng = np.random.default_rng(77)
    r = lambda *s: torch.from_numpy(rng.standard_normal(s, dtype=np.float32))
    out = {}
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        for name, (rt, rp) in {"eq": (6, 6), "fewer_pred": (6, 4), "fewer_tgt": (4, 6)}.items():
            t, p = r(rt, 32).to(dt), r(rp, 32).to(dt)
            out[f"l1_{name}_t_{tag}"], out[f"l1_{name}_p_{tag}"] = t, p
            out[f"l1_{name}_loss_{tag}"] = mse_loss_fn(t, p).float()
    save_npz("ops_r3.npz", **out)

    shared = np.random.default_rng(4177)
    for kind, extra_cfg, ctor in (("image_embeds", {}, {}), ("pretraining_tp2", {}, {})):
        cfg = tiny_cfg(num_image_tokens=4, **extra_cfg)
        sd = init_state_dict(cfg, seed=43)
        ids, lab = e2e_batch("mixed")
        ids_t, lab_t = torch.tensor(pad_rows(ids, 128001)), torch.tensor(pad_rows(lab, -100))
        msk_t = ids_t.ne(128001)
        n_img = sum(max(1, sum(1 for t in r_ if t == IM)) for r_ in ids)
        images = torch.from_numpy(shared.standard_normal((n_img, 3, 56, 56), dtype=np.float32))
        embeds = F.normalize(torch.from_numpy(shared.standard_normal((n_img, 4, cfg.v_hidden), dtype=np.float32)), dim=-1)
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            model = build_reference(cfg, sd, dt)
            if kind == "pretraining_tp2":
                model.config.pretraining_tp = 2
            for n, p in model.named_parameters():
                p.requires_grad_("vision_tower" not in n)
            if kind == "image_embeds":
                outp = model(input_ids=ids_t, attention_mask=msk_t, labels=lab_t, image_embeds=embeds.to(dt))
                tf = model.prepare_inputs_labels_for_multimodal(ids_t, None, msk_t, None, lab_t, None, image_embeds=embeds.to(dt))[7]
            else:
                outp = model(input_ids=ids_t, attention_mask=msk_t, labels=lab_t, images=images.to(dt))
                tf = model.prepare_inputs_labels_for_multimodal(ids_t, None, msk_t, None, lab_t, images.to(dt))[7]
            rec = dict(input_ids=ids_t, labels=lab_t, attention_mask=msk_t, images=images, image_embeds=embeds, seed=np.int64(43),
                       rows_per_image=np.int64(4), pretraining_tp=np.int64(getattr(model.config, "pretraining_tp", 1)),
                       target_features=tf.float(), loss=outp.loss.detach().float(), loss_language=np.float64(model.loss_language),
                       loss_image_ar=np.float64(model.loss_image_ar), logits_sub=outp.logits[:, :, ::997], hidden=outp.hidden_states)
            outp.loss.backward()
            for n, p in model.named_parameters():
                if p.grad is not None and "vision_proj" not in n:
                    rec["grad::" + n] = grad_summary(p.grad)
            save_npz(f"r3_{kind}_{tag}.npz", **rec)
            print(f"    {kind} {tag}: loss {float(outp.loss):.6f} lang {model.loss_language:.6f} img {model.loss_image_ar:.6f}")


def gen_hfgen(only=None):
    """Row "HF generate": the reference's `generate(use_customize_greedy=False)` (metamorph_llama.py:711-717) -> transformers
    GenerationMixin driving the reference's forward with a KV cache.  Weights as in gen_decode (sparse lm_head whose planned rows are
    solved on the hidden rows the reference itself produces, so every decision has a margin far above bf16 noise); recorded: the
    greedy ids, and a SAMPLING run (do_sample, temperature 0.7, top_p 0.9) -- the planned token holds > 0.9 of the probability mass at
    every step, so the nucleus is that one token and the sampled ids are seed-independent (checked with three seeds); and a beam search (num_beams = 2, both hypotheses returned with their scores)."""
    A = 128000
    rope31 = {"rope_scaling": ROPE31_TINY, "max_position_embeddings": 256}      # round 6: HF generate + KV cache under LLaMA-3.1's RoPE
    cases = {"text": ([[A, A, 11, 12, 13, 14, 15]], 0, {}), "image_prompt": ([[A, A, 11, ST, IM, EN, 12, 13]], 1, {}),
             "text_rope31": ([[A, A, 11, 12, 13, 14, 15]], 0, rope31)}
    plan = [(0, 41), (1, 42), (2, ST), (3, 43), (4, 44), (5, 128009)]      # HF path: <image_start> is an ordinary token, no image mode
    for ci, (name, (ids, n_img, extra)) in enumerate(cases.items()):
        if only is not None and name not in only:
            continue
        ids_t = torch.tensor(ids)
        seed = 71 + ci
        cfg = tiny_cfg(num_image_tokens=4, **extra)
        sd = init_state_dict(cfg, seed=seed)
        images = None
        if n_img:
            images = torch.from_numpy(np.random.default_rng(seed).standard_normal((n_img, 3, 56, 56), dtype=np.float32))

        def run(rows, dt, max_new=10, **gen_kw):
            sd2 = dict(sd)
            sd2["lm_head.weight"] = decode_lm_head(sd, rows)
            model = build_reference(cfg, sd2, dt)
            model.eval()
            lm_in, logits = [], []
            model.lm_head.register_forward_pre_hook(lambda m, a: lm_in.append(a[0][0, -1].detach().float().clone()))
            model.lm_head.register_forward_hook(lambda m, a, o: logits.append(o[0, -1].detach().float().clone()))
            with torch.no_grad():
                out = model.generate(inputs=ids_t, images=None if images is None else images.to(dt), use_customize_greedy=False,
                                     max_new_tokens=max_new, eos_token_id=128009, pad_token_id=128001, **gen_kw)
            return out[0].tolist(), lm_in, logits

        def solve(hid):
            steps = sorted(hid)
            H = torch.stack([hid[t] for t in steps]).double()
            pinv = torch.linalg.pinv(H)
            rows = {}
            for t, tok in plan:
                if t not in hid:
                    continue
                y = torch.tensor([14.0 if u == t else 0.0 for u in steps], dtype=torch.float64)
                rows[tok] = (pinv @ y).float()
            return rows

        hid, rows = {}, {}
        want = dict(plan)
        for _ in range(len(plan) + 1):
            toks, lm_in, logits = run(rows, torch.float32, do_sample=False)
            argm = [int(l.argmax()) for l in logits]
            good = 0
            for t in range(len(lm_in)):
                hid[t] = lm_in[t]
                if t in want and argm[t] != want[t]:
                    break
                good = t + 1
            if good >= plan[-1][0] + 1:
                break
            hid = {t: v for t, v in hid.items() if t <= good}
            rows = solve(hid)
        toks, lm_in, logits = run(rows, torch.float32, do_sample=False)
        assert toks == [tok for _, tok in plan], (name, toks)
        top2 = [l.topk(2) for l in logits]
        margins = torch.stack([t.values[0] - t.values[1] for t in top2])
        probs = torch.stack([torch.softmax(l / 0.7, -1).max() for l in logits])
        assert float(margins.min()) > 6.0 and float(probs.min()) > 0.95, (margins, probs)
        toks16, _, logits16 = run(rows, torch.bfloat16, do_sample=False)
        assert toks16 == toks
        sampled = []
        for sseed in (0, 1, 2):
            torch.manual_seed(sseed)
            sampled.append(run(rows, torch.float32, do_sample=True, temperature=0.7, top_p=0.9)[0])
        assert all(sm == toks for sm in sampled), sampled
        # beam search (num_beams = 2, both hypotheses returned): the second beam leaves the planned path, so the cache rows are re-ordered
        sd2 = dict(sd)
        sd2["lm_head.weight"] = decode_lm_head(sd, rows)
        beams = {}
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            model = build_reference(cfg, sd2, dt)
            model.eval()
            with torch.no_grad():
                bo = model.generate(inputs=ids_t, images=None if images is None else images.to(dt), use_customize_greedy=False, num_beams=2,
                                    num_return_sequences=2, do_sample=False, max_new_tokens=10, eos_token_id=128009, pad_token_id=128001,
                                    return_dict_in_generate=True, output_scores=True)
            beams[tag] = (bo.sequences.clone(), bo.sequences_scores.float().clone())
        assert beams["f32"][0].tolist() == beams["bf16"][0].tolist(), (beams["f32"][0].tolist(), beams["bf16"][0].tolist())
        assert beams["f32"][0][0].tolist()[:len(toks)] == toks
        # a BATCH of two prompts, the shorter one left-padded, with its attention mask (text case): HF derives position_ids from the mask, so
        # each row generates what it generates alone
        batch = {}
        if images is None:
            model = build_reference(cfg, sd2, torch.float32)
            model.eval()
            short = ids[0][:2] + ids[0][4:]
            rows2 = torch.tensor([ids[0], [128001] * (len(ids[0]) - len(short)) + short])
            with torch.no_grad():
                b_out = model.generate(inputs=rows2, images=None, attention_mask=rows2.ne(128001), use_customize_greedy=False, do_sample=False,
                                       max_new_tokens=6, eos_token_id=128009, pad_token_id=128001)
                alone = model.generate(inputs=torch.tensor([short]), images=None, use_customize_greedy=False, do_sample=False,
                                       max_new_tokens=6, eos_token_id=128009, pad_token_id=128001)
            batch = dict(batch_input_ids=rows2, batch_attention_mask=rows2.ne(128001), batch_sequences=b_out, short_alone_sequence=alone[0])
            print(f"    {name}: batched left-padded generate {b_out.tolist()}; the short prompt alone {alone[0].tolist()}")
        print(f"    {name}: beams {beams['f32'][0].tolist()} scores {beams['f32'][1].tolist()} (bf16 {beams['bf16'][1].tolist()})")
        save_npz(f"hfgen_{name}.npz", beam_sequences=beams["f32"][0], beam_scores=beams["f32"][1], beam_scores_bf16=beams["bf16"][1],
                 seed=np.int64(seed), input_ids=ids_t, images=images if images is not None else torch.zeros(0),
                 active=np.array(DECODE_ACTIVE, dtype=np.int64), row_tokens=np.array(list(rows.keys()), dtype=np.int64),
                 row_values=torch.stack(list(rows.values())), tokens=np.array(toks, dtype=np.int64), margins=margins,
                 top_prob_at_T07=probs, sampled_tokens=np.array(sampled[0], dtype=np.int64),
                 active_logits=torch.stack([l[DECODE_ACTIVE] for l in logits]),
                 active_logits_bf16=torch.stack([l[DECODE_ACTIVE] for l in logits16]), max_new_tokens=np.int64(10), **batch,
                 **({"cfg_json": np.array(json.dumps(extra))} if extra else {}))
        print(f"    {name}: seed {seed} tokens {toks} min margin {float(margins.min()):.2f} min top prob at T=0.7 {float(probs.min()):.4f}")


def gen_surface():
    """The rest of the Python surface SURVEY.md 8(b) lists, recorded from the reference: `get_model_name_from_path` (mm_utils.py:218-224),
    `KeywordsStoppingCriteria` (mm_utils.py:226-258: id-suffix match, decoded-text match over the last `max_keyword_len` NEW tokens, all rows
    of a batch must stop), `initialize_vision_tokenizer` (metamorph_arch.py:427-469: added tokens, embedding resize, new rows = mean of the
    old ones in the weights' own dtype, requires_grad policy, rows taken from a stage-1 adapter file of either shape)."""
    import tempfile
    from types import SimpleNamespace
    from metamorph.mm_utils import KeywordsStoppingCriteria, get_model_name_from_path
    from oracle.fake_tokenizer import VocabTokenizer
    out = {}
    paths = ["org/model-7b", "/abs/run/checkpoint-500", "runs/x/checkpoint-12/", "single", "/a/b/c/", "a/checkpoint-", "checkpoint-3",
             "x/y/checkpoints-3"]
    names = []
    for q in paths:
        try:
            names.append(get_model_name_from_path(q))
        except Exception as e:
            names.append(f"{type(e).__name__}")
    out["model_names"] = dict(zip(paths, names))

    tok = VocabTokenizer(add_bos=True)
    enc = lambda text: tok(text).input_ids[1:]
    prompt = torch.tensor([[128000] + enc("Human : the cat sat")])
    stop_cases = []
    for kws in (["###"], ["stop", "###"], ["answer is yes"], ["done ."]):
        for add_bos in (True, False):
            tk = VocabTokenizer(add_bos=add_bos)
            crit = KeywordsStoppingCriteria(kws, tk, prompt)
            for gen in ("", "a mat", "a mat ###", "### a", "answer is yes", "answer is yes .", "is yes", "the answer is yes no", "done", "done .",
                        "stop", "one two three four answer is yes", "answer is yes one two three four"):
                ids = torch.cat([prompt, torch.tensor([enc(gen)], dtype=torch.long)], 1)
                stop_cases.append(dict(keywords=kws, add_bos=add_bos, generated=gen, stop=bool(crit(ids, None)),
                                       max_keyword_len=int(crit.max_keyword_len), start_len=int(crit.start_len)))
    # batches: every row must stop
    crit = KeywordsStoppingCriteria(["###"], tok, prompt)
    for gens in (("a ###", "mat ###"), ("a ###", "mat a"), ("a a", "mat a")):
        ids = torch.cat([prompt.repeat(2, 1), torch.tensor([enc(g) for g in gens])], 1)
        stop_cases.append(dict(keywords=["###"], add_bos=True, generated=list(gens), stop=bool(crit(ids, None)), max_keyword_len=int(crit.max_keyword_len),
                               start_len=int(crit.start_len)))
    out["stopping"] = stop_cases
    out["prompt_ids"] = prompt[0].tolist()

    cfg = tiny_cfg(vocab_size=512, hidden_size=64, intermediate_size=128, num_image_tokens=4)
    sd = init_state_dict(cfg, seed=51)
    rng = np.random.default_rng(52)
    vt = []
    with tempfile.TemporaryDirectory() as tmp:
        adapters = {}
        for kind, rows in (("full", 514), ("rows", 2), ("bad", 7)):
            path = os.path.join(tmp, f"adapter_{kind}.bin")
            w = torch.from_numpy(rng.standard_normal((rows, cfg.hidden_size), dtype=np.float32))
            torch.save({"model.embed_tokens.weight": w}, path)
            adapters[kind] = (path, w)
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            for patch, se, tune, adapter in ((False, False, False, None), (True, False, True, None), (False, True, False, None), (False, True, True, None),
                                             (True, True, True, None), (False, True, False, "full"), (False, True, True, "rows"), (False, True, False, "bad")):
                model = build_reference(cfg, sd, dt)
                tk = VocabTokenizer()
                margs = SimpleNamespace(mm_use_im_patch_token=patch, mm_use_im_start_end=se, tune_mm_mlp_adapter=tune,
                                        pretrain_mm_mlp_adapter=adapters[adapter][0] if adapter else None)
                rec = dict(dtype=tag, mm_use_im_patch_token=patch, mm_use_im_start_end=se, tune_mm_mlp_adapter=tune, adapter=adapter)
                rg0 = (model.get_input_embeddings().weight.requires_grad, model.get_output_embeddings().weight.requires_grad)
                try:
                    model.initialize_vision_tokenizer(margs, tk)
                    rec["error"] = None
                except Exception as e:
                    rec["error"] = type(e).__name__
                ie, oe = model.get_input_embeddings().weight, model.get_output_embeddings().weight
                n_new = len(tk) - 512
                rec.update(len_tokenizer=len(tk), added=list(tk.added), embed_shape=list(ie.shape), lm_head_shape=list(oe.shape),
                           requires_grad_before=list(rg0), requires_grad=[ie.requires_grad, oe.requires_grad],
                           old_rows_unchanged=bool(torch.equal(ie.data[:512].float(), sd["model.embed_tokens.weight"].to(dt).float())
                                                   and torch.equal(oe.data[:512].float(), sd["lm_head.weight"].to(dt).float())),
                           config_vocab_size=int(model.config.vocab_size))
                # the start / end rows are deterministic (mean of the old rows, or adapter rows) unless a patch-token row -- random in the
                # reference too (HF's resize draws it) -- was appended first and entered that mean
                n_det = 2 if (se and n_new >= 2 and not patch) else 0
                rec["new_embed_rows"] = ie.data[ie.shape[0] - n_det:].float().tolist() if n_det else []
                rec["new_lm_head_rows"] = oe.data[oe.shape[0] - n_det:].float().tolist() if n_det else []
                vt.append(rec)
        # the test regenerates the adapter rows: default_rng(52).standard_normal((rows, hidden), float32) for rows = 514, 2, 7 in this order
        out["adapter_rows_sum"] = {k: float(v[1].double().sum()) for k, v in adapters.items()}
    out["vision_tokenizer"] = vt
    out["seed"] = 51
    # initialize_vision_modules (metamorph_arch.py:46-96): config side effects, projector un-frozen, vision_proj re-created, temperature_in
    vm = []
    for frozen, mtype in ((True, "mlp2x_gelu"), (False, "mlp2x_gelu")):
        model = build_reference(cfg, sd, torch.float32)
        inner = model.get_model()
        for p in inner.mm_projector.parameters():
            p.requires_grad_(not frozen)
        old_proj = inner.vision_proj
        t_before = inner.temperature_in
        margs = SimpleNamespace(vision_tower="siglip/CLIP-ViT-SO400M-14-384", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                                pretrain_mm_mlp_adapter=None, mm_projector_type=mtype, mm_patch_merge_type="flat")
        inner.vision_tower.load_model = lambda *a, **k: None      # the hub download cannot run offline; the tower is already built
        inner.initialize_vision_modules(margs, fsdp=None)
        c = model.config
        vm.append(dict(projector_frozen_before=frozen, temperature_in_before=t_before, temperature_in=inner.temperature_in,
                       vision_proj_is_new_object=inner.vision_proj is not old_proj, vision_proj_shape=list(inner.vision_proj.weight.shape),
                       projector_requires_grad=[p.requires_grad for p in inner.mm_projector.parameters()],
                       config={k: getattr(c, k) for k in ("mm_vision_tower", "use_mm_proj", "mm_projector_type", "mm_hidden_size", "mm_vision_select_layer",
                                                          "mm_vision_select_feature", "mm_patch_merge_type")},
                       tower_select_layer_attr=inner.vision_tower.select_layer))
    out["vision_modules"] = vm
    with open(os.path.join(OUT, "surface.json"), "w") as f:
        json.dump(out, f)
    print(f"  wrote surface.json: {len(names)} names, {len(stop_cases)} stopping cases ({sum(c['stop'] for c in stop_cases)} stop), "
          f"{len(vt)} tokenizer cases; errors {[c['error'] for c in vt if c['error']]}")


def gen_textonly():
    """forward(images=None): `prepare_inputs_labels_for_multimodal` returns early (metamorph_arch.py:184-191), the decoder embeds the ids itself,
    `image_positions` is None so the image-AR block is skipped (metamorph_llama.py:333, 420): loss = CE alone, `loss_language` / `loss_image_ar`
    are not touched.  Right-padded batch of two rows, labels on the tail of each."""
    cfg = tiny_cfg(num_image_tokens=4)
    sd = init_state_dict(cfg, seed=61)
    rng = np.random.default_rng(62)
    rows = [[128000] + rng.integers(3, 127000, 11).tolist(), [128000] + rng.integers(3, 127000, 6).tolist()]
    labs = [[-100] * 5 + rows[0][5:], [-100] * 3 + rows[1][3:]]
    ids_t, lab_t = torch.tensor(pad_rows(rows, 128001)), torch.tensor(pad_rows(labs, -100))
    msk_t = ids_t.ne(128001)
    _textonly_record(cfg, sd, ids_t, lab_t, msk_t, "r3_textonly")
    # the same batch LEFT-padded (a tokenizer with padding_side = "left"): HF's decoder takes position_ids = arange(L), padding included
    L = ids_t.shape[1]
    left = lambda rws, v: torch.tensor([[v] * (L - len(r)) + r for r in rws])
    cfg_l = tiny_cfg(num_image_tokens=4, tokenizer_padding_side="left")
    ids_l, lab_l = left(rows, 128001), left(labs, -100)
    _textonly_record(cfg_l, sd, ids_l, lab_l, ids_l.ne(128001), "r3_textonly_left")


def _textonly_record(cfg, sd, ids_t, lab_t, msk_t, stem):
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        model = build_reference(cfg, sd, dt)
        for n, p in model.named_parameters():
            p.requires_grad_("vision_tower" not in n)
        out = model(input_ids=ids_t, attention_mask=msk_t, labels=lab_t, images=None)
        rec = dict(input_ids=ids_t, labels=lab_t, attention_mask=msk_t, seed=np.int64(61), loss=out.loss.detach().float(),
                   logits_sub=out.logits[:, :, ::997], hidden=out.hidden_states, logits_shape=np.array(out.logits.shape),
                   has_loss_language=np.int64(hasattr(model, "loss_language")))
        out.loss.backward()
        for n, p in model.named_parameters():
            if p.grad is not None:
                rec["grad::" + n] = grad_summary(p.grad)
        rec["params_without_grad"] = np.array(sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None))
        save_npz(f"{stem}_{tag}.npz", **rec)
        print(f"    {tag}: loss {float(out.loss):.6f}, {sum(k.startswith('grad::') for k in rec)} gradients, no gradient for {rec['params_without_grad'].tolist()}, "
              f"has loss_language attr: {hasattr(model, 'loss_language')}")


def gen_optgroups():
    """`MetaMorphTrainer.create_optimizer` (metamorph_trainer.py:154-245) run through a stub trainer: which parameter NAMES land in which
    group (decay x {base lr, mm_projector_lr | vision_lr}), with the tower frozen and trainable.  `get_optimizer_cls_and_kwargs` is replaced by
    plain torch AdamW (it needs real TrainingArguments); the grouping under test happens before that call."""
    from types import SimpleNamespace
    T, MT = _import_train()
    import transformers
    # transformers==4.45.0 (the reference's pin, pyproject.toml:16) registers LlamaRMSNorm as a layer-norm type when modeling_llama is imported
    # (`ALL_LAYERNORM_LAYERS.append(LlamaRMSNorm)`), so RMSNorm weights get NO weight decay under the reference's own stack; the 5.x installed
    # here dropped that line.  Re-create the pinned behaviour for this recording (and say so in the fixture).
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    if LlamaRMSNorm not in MT.ALL_LAYERNORM_LAYERS:
        MT.ALL_LAYERNORM_LAYERS.append(LlamaRMSNorm)
    cfg = tiny_cfg(num_image_tokens=4)
    sd = init_state_dict(cfg, seed=47)
    out = []
    orig = transformers.Trainer.get_optimizer_cls_and_kwargs
    transformers.Trainer.get_optimizer_cls_and_kwargs = staticmethod(lambda args, model=None: (torch.optim.AdamW, dict(lr=1e-3)))
    try:
        for train_tower in (False, True):
            for mm_lr, v_lr in ((None, None), (2e-5, None), (None, 2e-6), (2e-5, 2e-6)):
                model = build_reference(cfg, sd, torch.float32)
                for n, p in model.named_parameters():
                    p.requires_grad_(train_tower or "vision_tower" not in n)
                name_of = {id(p): n for n, p in model.named_parameters()}
                stub = SimpleNamespace(model=model, optimizer=None, args=SimpleNamespace(weight_decay=0.05, mm_projector_lr=mm_lr, vision_lr=v_lr))
                opt = MT.MetaMorphTrainer.create_optimizer(stub)
                groups = [dict(weight_decay=g["weight_decay"], lr=g["lr"], names=[name_of[id(p)] for p in g["params"]]) for g in opt.param_groups]
                out.append(dict(train_tower=train_tower, mm_projector_lr=mm_lr, vision_lr=v_lr, base_lr=1e-3, groups=groups))
    finally:
        transformers.Trainer.get_optimizer_cls_and_kwargs = orig
    with open(os.path.join(OUT, "n4_optimizer_groups.json"), "w") as f:
        json.dump({"note": "grouping by the reference's create_optimizer with LlamaRMSNorm registered in ALL_LAYERNORM_LAYERS as transformers==4.45.0 "
                           "(the reference's pin) does at import", "cases": out}, f)
    for c in out:
        print("   ", c["train_tower"], c["mm_projector_lr"], c["vision_lr"], [(g["weight_decay"], g["lr"], len(g["names"])) for g in c["groups"]])


# ----------------------------------------------------------------------------- round 6: LLaMA-3.1 RoPE (rope_type "llama3")

ROPE_TABLE_CASES = {
    # name: (head_dim, theta, rope_scaling, max_position_embeddings)
    "llama31_8b": (128, 500000.0, ROPE31, 131072),                                  # meta-llama/Llama-3.1-8B config.json
    "llama32_1b": (64, 500000.0, dict(ROPE31, factor=32.0), 131072),               # Llama-3.2-1B: d = 64, factor 32
    "tiny_ctx32": (128, 500000.0, ROPE31_TINY, 256),                                # the e2e / decode fixtures' shrunk context
    "linear4": (128, 500000.0, {"rope_type": "linear", "factor": 4.0}, 8192),
    "default": (128, 500000.0, None, 8192),
}


def gen_rope31():
    """(a) r6_rope_tables.npz: the `inv_freq` buffer and cos / sin rows HF's own LlamaRotaryEmbedding -- the module the reference's decoder
    runs (metamorph_llama.py:349-359) -- produces for LLaMA-3.1 / 3.2 / linear / default configs, at positions up to 4095.
    (b) r6_rope31_long_{f32,bf16}.npz: the reference's forward + backward on ONE 4096-row sample (image + text) under the REAL LLaMA-3.1
    constants, where the stretched bands turn by up to ~pi less than the default RoPE's: loss, every 16th hidden row, gradient summaries --
    plus how far the same weights land under the default RoPE (the fixture's discriminating power)."""
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    pos = torch.tensor(sorted(set(list(range(0, 4096, 41)) + [1, 2, 3, 31, 32, 33, 2047, 2048, 4095])))
    rec = {"positions": pos}
    for name, (d, theta, rs, mp) in ROPE_TABLE_CASES.items():
        c = LlamaConfig(hidden_size=d * 2, num_attention_heads=2, rope_theta=theta, max_position_embeddings=mp, **({"rope_scaling": dict(rs)} if rs else {}))
        emb = LlamaRotaryEmbedding(c)
        rec[f"{name}::inv_freq"] = emb.inv_freq.clone()
        rec[f"{name}::attention_scaling"] = np.float64(emb.attention_scaling)
        rec[f"{name}::cfg"] = np.array(json.dumps(dict(head_dim=d, rope_theta=theta, rope_scaling=rs, max_position_embeddings=mp)))
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            cos, sin = emb(torch.zeros(1, 1, dtype=dt), pos[None])
            rec[f"{name}::cos_{tag}"], rec[f"{name}::sin_{tag}"] = cos[0], sin[0]
    save_npz("r6_rope_tables.npz", **rec)

    L, T = 4096, 4
    extra = {"rope_scaling": ROPE31, "max_position_embeddings": 131072}
    cfg = tiny_cfg(num_image_tokens=T, tokenizer_model_max_length=L, **extra)
    sd = init_state_dict(cfg, seed=67)
    # N(0, 0.02) projections give attention logits of std ~0.1: softmax is then flat over 4096 keys and NO positional encoding matters.
    # q_proj / k_proj x 5.5 puts the logits at std ~3, so which keys a query favours depends on the rotation
    QK_GAIN = 5.5
    for k in sd:
        if k.endswith("q_proj.weight") or k.endswith("k_proj.weight"):
            if "vision_tower" not in k:
                sd[k] = sd[k] * QK_GAIN
    rng = np.random.default_rng(6700)
    n_text = L - (T + 2) - 3
    text = rng.integers(0, 127999, size=n_text).tolist()
    ids = [[128000, 128000] + text + [ST, IM, EN, 128009]]                         # spliced length = 2 + n_text + 2 + T + 1 = L: an ANSWER image at the far end
    lab = [[-100] * len(ids[0])]
    for j in list(range(300, 364)) + list(range(len(ids[0]) - 64, len(ids[0]))):  # 64 live targets early, 64 at the far end (incl. the image span)
        lab[0][j] = ids[0][j]
    ids_t, lab_t = torch.tensor(ids), torch.tensor(lab)
    msk_t = torch.ones_like(ids_t, dtype=torch.bool)
    images = torch.from_numpy(rng.standard_normal((1, 3, 56, 56), dtype=np.float32))
    rows = torch.arange(0, L, 16)
    hid_default = None
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        model = build_reference(cfg, sd, dt)
        for n, p in model.named_parameters():
            p.requires_grad_("vision_tower" not in n)
        out = model(input_ids=ids_t, attention_mask=msk_t, labels=lab_t, images=images.to(dt))
        assert out.hidden_states.shape[1] == L, out.hidden_states.shape
        rec = dict(input_ids=ids_t, labels=lab_t, attention_mask=msk_t, images=images, seed=np.int64(67), rows_per_image=np.int64(T),
                   cfg_json=np.array(json.dumps(dict(extra, tokenizer_model_max_length=L))), hidden_rows=rows, qk_gain=np.float64(QK_GAIN),
                   loss=out.loss.detach().float(), loss_language=np.float64(model.loss_language), loss_image_ar=np.float64(model.loss_image_ar),
                   hidden=out.hidden_states[0, rows])
        out.loss.backward()
        for n, p in model.named_parameters():
            if p.grad is not None and "vision_proj" not in n:
                rec["grad::" + n] = grad_summary(p.grad)
        if dt == torch.float32:
            plain = build_reference(tiny_cfg(num_image_tokens=T, tokenizer_model_max_length=L), sd, dt)
            with torch.no_grad():
                o2 = plain(input_ids=ids_t, attention_mask=msk_t, labels=lab_t, images=images)
            h1, h0 = out.hidden_states[0, rows].detach(), o2.hidden_states[0, rows]
            far = rows >= 2048
            rec["default_rope_hidden_rel"] = np.float64(float((h0 - h1).norm() / h1.norm()))
            rec["default_rope_hidden_rel_far"] = np.float64(float((h0[far] - h1[far]).norm() / h1[far].norm()))
            rec["default_rope_loss"] = o2.loss.detach().float()
            print(f"    long: loss {float(out.loss):.6f}; under the DEFAULT RoPE the same weights give loss {float(o2.loss):.6f}, hidden rows differ by "
                  f"{rec['default_rope_hidden_rel']:.3e} (rows >= 2048: {rec['default_rope_hidden_rel_far']:.3e})")
            assert rec["default_rope_hidden_rel_far"] > 5e-2
        save_npz(f"r6_rope31_long_{tag}.npz", **rec)


def gen_r6():
    """Everything round 6 added, without re-running the older generators: the new e2e cases, the rope31 decode / HF-generate cases, tables."""
    first = next(i for i, c in enumerate(E2E_CASES) if len(c) > 6)
    gen_e2e(first=first)
    gen_decode(only=("image_prompt_rope31",))
    gen_hfgen(only=("text_rope31",))
    gen_rope31()


def gen_r6tail():
    gen_decode(only=("image_prompt_rope31",))
    gen_hfgen(only=("text_rope31",))
    gen_rope31()



if __name__ == "__main__":
    # every generator, in an order that reproduces the committed fixtures bit for bit in ONE process (`optgroups` last: it registers
    # LlamaRMSNorm as a layer-norm type for the rest of the process); ~4 minutes on 8 threads
    ALL = ["a1", "a5", "a5rand", "a3", "a3sel", "ops", "e2e", "n2", "n2rand", "images", "conv", "decode", "names", "n3", "r3", "hfgen", "surface", "textonly",
           "rope31", "optgroups"]
    which = sys.argv[1:] or ALL
    for w in which:
        print(f"[gen_golden] {w}")
        globals()["gen_" + w]()
