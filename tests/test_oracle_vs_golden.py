"""Pins the CPU oracle (oracle/) against the golden vectors recorded from the reference itself
(oracle/gen_golden.py).  CPU only; runs in the `-m "not gpu"` tier."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops as ops
from oracle.fake_tokenizer import FakeTokenizer
from oracle.ref_model import OracleConfig, forward, init_state_dict, vision_features, mm_projector, siglip_hidden
from oracle.ref_plan import splice_bookkeeping, tokenizer_image_token

from conftest import GOLDEN


def T(a, dt=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dt) if dt is not None else t


def tiny_cfg(**kw):
    base = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=1, vocab_size=128258, v_layers=2, v_intermediate=144, v_image=56,
                num_image_tokens=4, tokenizer_model_max_length=64)
    base.update(kw)
    return OracleConfig(**base)


# ------------------------------------------------------------------ A1

def test_a1_tokenizer_image_token():
    cases = json.load(open(os.path.join(GOLDEN, "a1_tokenizer.json")))
    assert len(cases) >= 40
    for c in cases:
        tok = FakeTokenizer(add_bos=c["add_bos"])
        assert tokenizer_image_token(c["prompt"], tok, c["image_token_index"]) == c["ids"], c["prompt"]


# ------------------------------------------------------------------ A5

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "a5_*.npz"))))
def test_a5_splice_bookkeeping(path):
    g = np.load(path)
    Timg = int(g["rows_per_image"])
    labels = g["labels"].tolist() if ("labels_given" not in g.files or int(g["labels_given"])) else None       # a5_wrap_*: None handling
    mask = g["attention_mask"].tolist() if ("mask_given" not in g.files or int(g["mask_given"])) else None
    args = (g["input_ids"].tolist(), labels, mask, int(g["num_images"]), Timg, int(g["max_length"]) if int(g["max_length"]) >= 0 else None,
            "left" if int(g["left"]) else "right")
    if "error" in g.files:
        with pytest.raises(TypeError):
            splice_bookkeeping(*args)
        return
    plan = splice_bookkeeping(*args)
    src = [[-1 if s is None else (-2 - (s[1] * Timg + s[2]) if isinstance(s, tuple) else s) for s in row]
           for row in plan["src"]]
    assert np.array_equal(np.array(src), g["out_src"])
    if "out_labels" in g.files:
        assert np.array_equal(np.array(plan["labels"]), g["out_labels"])
    else:
        assert plan["labels"] is None
    if "out_attention_mask" in g.files:
        assert np.array_equal(np.array(plan["attention_mask"]), g["out_attention_mask"])
    assert np.array_equal(np.array(plan["image_positions"]), g["out_image_positions"])
    assert plan["target_keep"] == g["out_target_keep"].tolist()
    if "out_position_ids" in g.files:
        assert int(g["out_position_ids_is_none"]) == 0 and np.array_equal(np.array(plan["position_ids"]), g["out_position_ids"])
    else:
        assert int(g["out_position_ids_is_none"]) == 1


# ------------------------------------------------------------------ per-op

OPS = np.load(os.path.join(GOLDEN, "ops.npz"))
TOL = {"f32": dict(rtol=2e-5, atol=2e-6), "bf16": dict(rtol=1.6e-2, atol=1e-2)}
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_ops(tag):
    dt, tol = DT[tag], TOL[tag]
    g = lambda k: T(OPS[f"{k}_{tag}"])
    close = lambda a, b, **kw: torch.testing.assert_close(a.float(), b.float(), **{**tol, **kw})
    close(ops.rmsnorm(g("rms_x").to(dt), g("rms_w").to(dt), 1e-5), g("rms_y"))
    pos = torch.arange(9)[None].expand(2, 9)
    cos, sin = ops.rope_tables(pos, 128, 500000.0, dt)
    close(cos, g("rope_cos")); close(sin, g("rope_sin"))
    qe = ops.rope_apply(g("rope_q").to(dt), cos, sin)
    ke = ops.rope_apply(g("rope_k").to(dt), cos, sin)
    close(qe, g("rope_qe")); close(ke, g("rope_ke"))
    o = ops.attention(g("rope_qe").to(dt), g("rope_ke").to(dt), g("att_v").to(dt), T(OPS[f"att_valid_{tag}"]).bool())
    close(o, g("att_o"))
    close(ops.swiglu(g("swi_g").to(dt), g("swi_u").to(dt)), g("swi_y"))
    close(ops.gelu_erf(g("swi_g").to(dt).float()).to(dt), g("gelu_erf"))
    close(ops.gelu_tanh(g("swi_g").to(dt).float()).to(dt), g("gelu_tanh"))
    close(ops.layernorm(g("swi_g").to(dt), g("ln_w").to(dt), g("ln_b").to(dt), 1e-6), g("ln_y"))
    for a, b in ((27, 16), (27, 8), (4, 2)):
        x = g(f"interp_{a}_{b}_x").to(dt)
        y = ops.bilinear_reduce(x, b * b)
        close(y, g(f"interp_{a}_{b}_y"))
        close(ops.l2_normalize(y), g(f"interp_{a}_{b}_yn"))
    ce = ops.shifted_cross_entropy(g("ce_logits"), T(OPS[f"ce_labels_{tag}"]))
    close(ce, g("ce_loss"))
    close(ops.cosine_loss(g("cos_t").to(dt), g("cos_p").to(dt)), g("cos_loss"))


def test_adamw_matches_torch():
    p = T(OPS["adam_p0"]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(3):
        ops.adamw_step(p, T(OPS["adam_grads"][s]), m, v, s + 1, 1e-2, 0.9, 0.95, 1e-8, 0.1)
    torch.testing.assert_close(p, T(OPS["adam_p3"]), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ A3 tower

@pytest.mark.parametrize("Timg", [4, 16])
@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_a3_tower(Timg, tag):
    g = np.load(os.path.join(GOLDEN, f"a3_tower_T{Timg}_{tag}.npz"))
    cfg = tiny_cfg(num_image_tokens=Timg)
    dt = DT[tag]
    sd = init_state_dict(cfg, seed=int(g["seed"]), dtype=dt)
    images = T(g["images"]).to(dt)
    tol = dict(rtol=1e-4, atol=2e-5) if tag == "f32" else dict(rtol=5e-2, atol=3e-2)
    with torch.no_grad():
        raw = siglip_hidden(sd, cfg, images)
        torch.testing.assert_close(raw[:, :, ::8].float(), T(g["raw_hidden"]), **tol)
        feat = vision_features(sd, cfg, images)
        tolf = dict(rtol=1e-4, atol=1e-5) if tag == "f32" else dict(rtol=5e-2, atol=4e-3)
        torch.testing.assert_close(feat.float(), T(g["features"]), **tolf)
        proj = mm_projector(sd, cfg, feat)
        torch.testing.assert_close(proj[:, :, ::4].float(), T(g["projected"]), **tolf)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_a3_tower_select_layer_minus_2(tag):
    """`mm_vision_select_layer = -2` (siglip_encoder.py:129-131): hidden_states[-2] of a 3-layer tower = the oracle's tower cut after layer 2;
    `num_image_tokens = -1` returns zeros of the un-reduced shape (:146-148)."""
    g = np.load(os.path.join(GOLDEN, f"a3sel_tower_{tag}.npz"))
    dt = DT[tag]
    sd = init_state_dict(tiny_cfg(num_image_tokens=4, v_layers=3), seed=int(g["seed"]), dtype=dt)
    cfg = tiny_cfg(num_image_tokens=4, v_layers=2)               # layers 0 and 1 of the same state dict
    images = T(g["images"]).to(dt)
    tol = dict(rtol=1e-4, atol=2e-5) if tag == "f32" else dict(rtol=5e-2, atol=3e-2)
    with torch.no_grad():
        torch.testing.assert_close(siglip_hidden(sd, cfg, images)[:, :, ::8].float(), T(g["raw_hidden_m2"]), **tol)
        tolf = dict(rtol=1e-4, atol=1e-5) if tag == "f32" else dict(rtol=5e-2, atol=4e-3)
        torch.testing.assert_close(vision_features(sd, cfg, images).float(), T(g["features_m2"]), **tolf)
    assert float(g["tokens_minus1_absmax"]) == 0.0 and g["tokens_minus1_shape"].tolist() == [2, 16, 1152]


# ------------------------------------------------------------------ end to end

def head_variant(g):
    """normalize_vision / apply_softmax of an e2e fixture (A8: cosine, mean-abs and soft-CE heads)."""
    return dict(normalize_vision=bool(int(g["normalize_vision"])), apply_softmax=bool(int(g["apply_softmax"])),
                tokenizer_padding_side="left" if int(g["left"]) else "right", mm_projector_type=str(g["mm_projector_type"]),
                image_token_reduction=str(g["image_token_reduction"]),
                **({"vision_head_type": str(g["vision_head_type"])} if "vision_head_type" in g else {}),
                **({"vision_coef": float(g["vision_coef"])} if "vision_coef" in g else {}),
                **(json.loads(str(g["cfg_json"])) if "cfg_json" in g else {}))


def _grad_summary(t):
    f = t.detach().float().flatten()
    n = min(256, f.numel())
    idx = (torch.arange(n, dtype=torch.long) * (f.numel() - 1)) // max(n - 1, 1)
    return torch.cat([f.norm()[None], f[idx]])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "e2e_*_f32.npz"))))
def test_e2e_fp32(path):
    g = np.load(path)
    cfg = tiny_cfg(num_image_tokens=int(g["rows_per_image"]), use_vision_ar=bool(int(g["use_vision_ar"])), **head_variant(g))
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    for k, v in sd.items():
        if "vision_tower" not in k:
            v.requires_grad_(True)
    out = forward(sd, cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), T(g["images"]))
    ref_loss = float(g["loss"])
    if np.isnan(ref_loss):
        assert torch.isnan(out["loss"])
    else:
        assert abs(float(out["loss"]) - ref_loss) < 2e-5 * max(1, abs(ref_loss))
    assert abs(out["loss_language"] - float(g["loss_language"])) < 2e-5 * max(1, abs(ref_loss))
    if np.isnan(float(g["loss_image_ar"])):
        assert np.isnan(out["loss_image_ar"])            # the A9 quirk: no answer-side image -> NaN
    else:
        assert abs(out["loss_image_ar"] - float(g["loss_image_ar"])) < 2e-5
    # left padding: a padding row sees no key at all (fully masked softmax row: implementation-defined garbage in the reference stack) --
    # outside the contract (labels -100, image_positions 0 there); every other case is compared on all rows
    rows = out["attention_mask"] if int(g["left"]) else torch.ones_like(out["attention_mask"])
    torch.testing.assert_close(out["logits"][:, :, ::997][rows], T(g["logits_sub"])[rows], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out["hidden_states"][rows], T(g["hidden"])[rows], rtol=1e-4, atol=1e-5)
    if torch.isfinite(out["loss"]):
        out["loss"].backward()
        n = 0
        for k in g.files:
            if k.startswith("grad::"):
                name = k[6:]
                assert sd[name].grad is not None, name
                torch.testing.assert_close(_grad_summary(sd[name].grad), T(g[k]), rtol=2e-4, atol=1e-6)
                n += 1
        assert n >= 20


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "e2e_*_bf16.npz"))))
def test_e2e_bf16(path):
    g = np.load(path)
    cfg = tiny_cfg(num_image_tokens=int(g["rows_per_image"]), use_vision_ar=bool(int(g["use_vision_ar"])), **head_variant(g))
    sd = init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16)
    with torch.no_grad():
        out = forward(sd, cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), T(g["images"]).bfloat16())
    ref_loss = float(g["loss"])
    if np.isnan(ref_loss):
        assert torch.isnan(out["loss"])
    else:
        # north_star tolerance: 1e-3 in bf16 (relative, on the loss)
        assert abs(float(out["loss"]) - ref_loss) < 1e-3 * max(1, abs(ref_loss)) * 3
    rows = out["attention_mask"] if int(g["left"]) else torch.ones_like(out["attention_mask"])
    torch.testing.assert_close(out["hidden_states"].float()[rows], T(g["hidden"])[rows], rtol=5e-2, atol=5e-2)


# ------------------------------------------------------------------ N1: the greedy decode loop

@pytest.mark.parametrize("name", ["text", "image_prompt", "image_prompt_rope31"])
def test_n1_greedy_decode_matches_reference_loop(name):
    from oracle.ref_model import decode_fixture_state_dict, greedy_decode
    g = np.load(os.path.join(GOLDEN, f"n1_decode_{name}.npz"))
    cfg = tiny_cfg(num_image_tokens=4, **(json.loads(str(g["cfg_json"])) if "cfg_json" in g else {}))
    sd = decode_fixture_state_dict(g, cfg)
    images = T(g["images"]) if g["images"].size else None
    toks, pz, logits = greedy_decode(sd, cfg, T(g["input_ids"]), images, max_new_tokens=int(g["max_new_tokens"]))
    assert toks == g["tokens"].tolist()
    assert [int(l.argmax()) for l in logits] == g["step_argmax"].tolist()
    torch.testing.assert_close(pz, T(g["pred_z"]), rtol=1e-4, atol=1e-6)
    act = g["active"].tolist()
    torch.testing.assert_close(torch.stack([l[act] for l in logits]), T(g["active_logits"]), rtol=1e-4, atol=2e-5)
    for mn in (2, 6):                                          # max_new_tokens counts image-mode iterations too (:587-590)
        t_m, pz_m, l_m = greedy_decode(sd, cfg, T(g["input_ids"]), images, max_new_tokens=mn)
        assert (t_m, pz_m.shape[0], len(l_m)) == (g[f"tokens_max{mn}"].tolist(), int(g[f"n_pred_z_max{mn}"]), int(g[f"iterations_max{mn}"]))


def test_mean_abs_loss_row_mismatch_matches_reference():
    """`mse_loss_fn` of the reference with equal / fewer prediction / fewer target rows (oracle/gen_golden.py r3 -> ops_r3.npz)."""
    g = np.load(os.path.join(GOLDEN, "ops_r3.npz"))
    for name in ("eq", "fewer_pred", "fewer_tgt"):
        t, p = torch.from_numpy(g[f"l1_{name}_t_f32"]), torch.from_numpy(g[f"l1_{name}_p_f32"])
        want = float(g[f"l1_{name}_loss_f32"])
        assert abs(float(ops.mean_abs_loss(t, p)) - want) <= 2e-6 * abs(want), name
        # the reference's bf16 run accumulates the per-row means in bf16: three significant digits
        t16, p16 = torch.from_numpy(g[f"l1_{name}_t_bf16"]), torch.from_numpy(g[f"l1_{name}_p_bf16"])
        assert abs(float(ops.mean_abs_loss(t16, p16)) - float(g[f"l1_{name}_loss_bf16"])) <= 1e-2 * abs(want), name


@pytest.mark.parametrize("kind", ["image_embeds", "pretraining_tp2"])
def test_oracle_image_embeds_and_pretraining_tp_match_reference(kind):
    """forward(image_embeds=...) -> `encode_imagesembed` (metamorph_arch.py:166-173) and `pretraining_tp = 2` (metamorph_llama.py:393-396)
    recorded from the reference (oracle/gen_golden.py r3)."""
    g = np.load(os.path.join(GOLDEN, f"r3_{kind}_f32.npz"))
    cfg = OracleConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                       vocab_size=128258, v_layers=2, v_intermediate=144, v_image=56, num_image_tokens=4, tokenizer_model_max_length=64,
                       pretraining_tp=int(g["pretraining_tp"]))
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    for k, v in sd.items():
        v.requires_grad_("vision_tower" not in k and "vision_proj" not in k)
    T = lambda k: torch.from_numpy(g[k])
    out = forward(sd, cfg, T("input_ids"), T("attention_mask"), T("labels"), T("images"),
                  image_embeds=T("image_embeds") if kind == "image_embeds" else None)
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    assert abs(out["loss_language"] - float(g["loss_language"])) <= 2e-5 * abs(float(g["loss_language"]))
    assert abs(out["loss_image_ar"] - float(g["loss_image_ar"])) <= 2e-5
    assert torch.allclose(out["target_features"], T("target_features"), atol=1e-6)
    assert torch.allclose(out["logits"][:, :, ::997].detach(), T("logits_sub"), atol=2e-4, rtol=2e-4)
    assert torch.allclose(out["hidden_states"].detach(), T("hidden"), atol=2e-4, rtol=2e-4)
    out["loss"].backward()
    n = 0
    for k in g.files:
        if k.startswith("grad::"):
            gr = sd[k[6:]].grad
            f = gr.flatten()
            m = min(256, f.numel())
            idx = (torch.arange(m, dtype=torch.long) * (f.numel() - 1)) // max(m - 1, 1)
            got = torch.cat([f.norm()[None], f[idx]])
            assert torch.allclose(got, T(k), atol=1e-6, rtol=2e-3), k
            n += 1
    assert n >= 25


@pytest.mark.parametrize("stem", ["r3_textonly", "r3_textonly_left"])
@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_text_only_forward_without_images(tag, stem):
    """forward(images=None) as the reference ran it (r3_textonly_*.npz): early return of the splice, CE alone, no gradient for the projector /
    vision head, `loss_language` never set."""
    g = np.load(os.path.join(GOLDEN, f"{stem}_{tag}.npz"))
    cfg = tiny_cfg(num_image_tokens=4, tokenizer_padding_side="left" if stem.endswith("left") else "right")
    dt = DT[tag]
    sd = init_state_dict(cfg, seed=int(g["seed"]), dtype=dt)
    for k, v in sd.items():
        v.requires_grad_("vision_tower" not in k)
    out = forward(sd, cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), None)
    tol = 2e-5 if tag == "f32" else 5e-3
    assert abs(float(out["loss"]) - float(g["loss"])) < tol * float(g["loss"])
    assert list(out["logits"].shape) == g["logits_shape"].tolist() and "loss_language" not in out and int(g["has_loss_language"]) == 0
    valid = T(g["attention_mask"]).bool()
    # padded rows: the reference's SDPA output there depends on its mask fill value; the loss never reads them
    torch.testing.assert_close(out["hidden_states"].float()[valid], T(g["hidden"])[valid], rtol=1e-4 if tag == "f32" else 5e-2, atol=2e-5 if tag == "f32" else 5e-2)
    torch.testing.assert_close(out["logits"][:, :, ::997].float()[valid], T(g["logits_sub"])[valid], rtol=1e-4 if tag == "f32" else 5e-2, atol=5e-5 if tag == "f32" else 5e-2)
    out["loss"].backward()
    no_grad = sorted(k for k, v in sd.items() if v.requires_grad and v.grad is None)
    assert no_grad == g["params_without_grad"].tolist()
    if tag == "f32":
        for k in g.files:
            if k.startswith("grad::"):
                torch.testing.assert_close(_grad_summary(sd[k[6:]].grad), T(g[k]), rtol=2e-4, atol=2e-6)


# ------------------------------------------------------------------ round 6: LLaMA-3.1 RoPE (rope_type "llama3") and friends

ROPE_TABLES = np.load(os.path.join(GOLDEN, "r6_rope_tables.npz"))
ROPE_CASES = sorted({k.split("::")[0] for k in ROPE_TABLES.files if "::" in k})


@pytest.mark.parametrize("name", ROPE_CASES)
def test_rope_inv_freq_and_tables_match_hf_buffers(name):
    """oracle.ref_ops.rope_inv_freq / rope_tables AND the product's host step (metamorph_amd.rope.rope_params -- pure host code, no kernel)
    against the buffers HF's LlamaRotaryEmbedding built for the same config: inv_freq bit for bit, cos / sin rows at positions <= 4095."""
    from types import SimpleNamespace
    from metamorph_amd.rope import rope_params
    g = ROPE_TABLES
    c = json.loads(str(g[f"{name}::cfg"]))
    want = T(g[f"{name}::inv_freq"])
    got = ops.rope_inv_freq(c["head_dim"], c["rope_theta"], c["rope_scaling"], c["max_position_embeddings"])
    assert torch.equal(got, want), float((got - want).abs().max())
    hf_like = SimpleNamespace(hidden_size=2 * c["head_dim"], num_attention_heads=2, rope_theta=c["rope_theta"], rope_scaling=c["rope_scaling"],
                              max_position_embeddings=c["max_position_embeddings"])
    rp = rope_params(hf_like)
    assert np.array_equal(rp.inv_freq, g[f"{name}::inv_freq"]) and rp.attention_scaling == float(g[f"{name}::attention_scaling"]) == 1.0
    assert rp.head_dim == c["head_dim"] and rp.rope_type == ((c["rope_scaling"] or {}).get("rope_type", "default"))
    pos = T(g["positions"])
    for tag, dt in DT.items():
        cos, sin = ops.rope_tables(pos, c["head_dim"], c["rope_theta"], dt, c["rope_scaling"], c["max_position_embeddings"])
        assert torch.equal(cos.float(), T(g[f"{name}::cos_{tag}"])) and torch.equal(sin.float(), T(g[f"{name}::sin_{tag}"]))
    if name != "default":                                      # the scaled tables are not the default ones in disguise
        assert not np.array_equal(g[f"{name}::cos_bf16"], g["default::cos_bf16"])


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_rope31_long_sample_real_llama31_constants(tag):
    """One 4096-row sample under the REAL LLaMA-3.1 RoPE constants, recorded from the reference (oracle/gen_golden.py rope31): loss, every 16th
    hidden row, gradient summaries.  The fixture itself states how far the default RoPE lands on the same weights (hidden rows 0.75 apart)."""
    g = np.load(os.path.join(GOLDEN, f"r6_rope31_long_{tag}.npz"))
    g32 = np.load(os.path.join(GOLDEN, "r6_rope31_long_f32.npz"))
    assert float(g32["default_rope_hidden_rel_far"]) > 0.5
    cfg = tiny_cfg(num_image_tokens=int(g["rows_per_image"]), **json.loads(str(g["cfg_json"])))
    sd = long_rope_state_dict(cfg, g, DT[tag])
    if tag == "f32":
        for k, v in sd.items():
            if "vision_tower" not in k:
                v.requires_grad_(True)
    with torch.set_grad_enabled(tag == "f32"):
        out = forward(sd, cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), T(g["images"]).to(DT[tag]), return_logits=False,
                      ce_rows_only=True)
    rows = T(g["hidden_rows"])
    if tag == "f32":
        assert abs(float(out["loss"]) - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
        torch.testing.assert_close(out["hidden_states"][0, rows], T(g["hidden"]), rtol=2e-4, atol=2e-5)
        out["loss"].backward()
        n = 0
        for k in g.files:
            if k.startswith("grad::"):
                torch.testing.assert_close(_grad_summary(sd[k[6:]].grad), T(g[k]), rtol=5e-4, atol=2e-6)
                n += 1
        assert n >= 20
    else:
        # sharp attention over 4096 keys in bf16: the reference's own bf16 run sits 4.4e-3 (relative) off its fp32 run; the yardstick for
        # any bf16 implementation is the fp32 truth, at the reference-bf16 run's distance
        l32 = float(g32["loss"])
        assert abs(float(out["loss"]) - l32) <= max(1.5 * abs(float(g["loss"]) - l32), 1e-3 * abs(l32)), (float(out["loss"]), float(g["loss"]), l32)
        e = float((out["hidden_states"][0, rows].float() - T(g32["hidden"])).norm() / T(g32["hidden"]).norm())
        e_ref = float((T(g["hidden"]) - T(g32["hidden"])).norm() / T(g32["hidden"]).norm())
        assert e <= 1.5 * e_ref, (e, e_ref)


def long_rope_state_dict(cfg, g, dtype):
    """The rope31 long fixture's weights: seeded state dict with q_proj / k_proj x qk_gain (attention logits of std ~3, so positions matter)."""
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    gain = float(g["qk_gain"])
    for k in sd:
        if (k.endswith("q_proj.weight") or k.endswith("k_proj.weight")) and "vision_tower" not in k:
            sd[k] = sd[k] * gain
    return {k: v.to(dtype) for k, v in sd.items()}
