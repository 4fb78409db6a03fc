"""End-to-end parity of the HIP model (metamorph_amd.model) against (a) the golden vectors recorded from the
reference itself and (b) the CPU oracle on the same seeded weights.  Needs an MI355X:  pytest -m gpu

Tolerance (north_star: "within 1e-3 bf16 tolerance"): the reference's own bf16 run differs from its fp32 run by
a data-dependent amount; we require the HIP bf16 result to be as close to the fp32 truth as the reference's bf16
result is, up to a factor of 1.5, and every loss to agree with the reference-bf16 loss AND the fp32 truth to 1e-3 relative.
Floors next to the factors are 1.5 x the value measured on MI355X in round 3 (profiles/r3_parity_measured.log), so a 2 x
regression of any of them fails.  Integer outputs are compared bit-exactly.
"""
import glob
import json
import math
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle.ref_model import OracleConfig, forward as oracle_forward, init_state_dict  # noqa: E402

DEV = "cuda"

# ---- the full-width / full-depth tests (GPUTEST budget: the whole -m gpu suite must fit the driver's step limit)
# (1) their fp32 oracle is evaluated with stock torch fp32 ops ON THE GPU by default (oracle/ref_stream.full_depth(device=...): same
#     functions, seconds instead of minutes of host GEMMs; pinned to the reference's recorded fp32 run and to the host evaluation by
#     test_streamed_oracle_on_device_*).  MM355_ORACLE_FP32_DEVICE=cpu restores the host evaluation.
# (2) their bf16 YARDSTICK -- the oracle run in bf16 on the host, i.e. the reference stack's own bf16 arithmetic, whose distance from the
#     fp32 truth bounds what "within bf16 tolerance" means at that depth -- is a deterministic function of the seeded weights and inputs:
#     it is recorded once (MM355_RECORD_YARDSTICK=1 pytest ... -> gpurun_out/r6_bf16_yardstick.json) and committed as
#     tests/golden/r6_bf16_yardstick.json together with the fp32 oracle's loss on the same inputs, which every run re-checks (a changed
#     seed / batch / weight generator fails loudly instead of silently using a stale yardstick).  No entry for a key: the yardstick is
#     computed on the spot, as before.
ORACLE_FP32_DEVICE = os.environ.get("MM355_ORACLE_FP32_DEVICE", "cuda")
RECORD_YARDSTICK = os.environ.get("MM355_RECORD_YARDSTICK") == "1"
YARDSTICK_PATH = os.path.join(GOLDEN, "r6_bf16_yardstick.json")


def load_yardstick(key):
    if RECORD_YARDSTICK or not os.path.exists(YARDSTICK_PATH):
        return None
    with open(YARDSTICK_PATH) as f:
        return json.load(f).get(key)


def record_yardstick(key, rec):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "r6_bf16_yardstick.json")
    allr = json.load(open(path)) if os.path.exists(path) else {}
    allr[key] = rec
    with open(path, "w") as f:
        json.dump(allr, f, indent=1, sort_keys=True)


def oracle_fp32(fetch_host, fetch_dev, cfg, ids, mask, labels, images_f32, **kw):
    """The layer-streamed fp32 oracle on ORACLE_FP32_DEVICE (results always on the host).  fetch_host(k) -> fp32 host tensor,
    fetch_dev(k) -> fp32 device tensor of the same values."""
    from oracle.ref_stream import full_depth
    if ORACLE_FP32_DEVICE == "cpu" or RECORD_YARDSTICK:
        return full_depth(fetch_host, cfg, ids, mask, labels, images_f32, **kw)
    return full_depth(fetch_dev, cfg, ids, mask, labels, images_f32, device=ORACLE_FP32_DEVICE, **kw)


def T(a):
    return torch.from_numpy(np.asarray(a))


def tiny_cfg(**kw):
    base = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=1, vocab_size=128258, v_layers=2, v_intermediate=144, v_image=56,
                num_image_tokens=4, tokenizer_model_max_length=64)
    base.update(kw)
    return OracleConfig(**base)


def hip_model(cfg: OracleConfig, sd, **kw):
    from metamorph_amd.factory import build_model
    llm = dict(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
               num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
               vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
               max_position_embeddings=cfg.max_position_embeddings, tie_word_embeddings=cfg.tie_word_embeddings,
               **({"rope_scaling": dict(cfg.rope_scaling)} if cfg.rope_scaling else {}),
               **({"head_dim": cfg.head_dim_explicit} if cfg.head_dim_explicit else {}))
    geo = dict(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_intermediate, num_hidden_layers=cfg.v_layers,
               num_attention_heads=cfg.v_heads, image_size=cfg.v_image, patch_size=cfg.v_patch, layer_norm_eps=cfg.v_ln_eps)
    return build_model(llm, geo, num_image_tokens=cfg.num_image_tokens, use_vision_ar=cfg.use_vision_ar,
                       normalize_vision=cfg.normalize_vision, apply_softmax=cfg.apply_softmax, image_start_id=cfg.image_start_id,
                       mm_projector_type=cfg.mm_projector_type, image_token_reduction=cfg.image_token_reduction,
                       vision_coef=cfg.vision_coef, max_length=cfg.tokenizer_model_max_length,
                       padding_side=cfg.tokenizer_padding_side, state_dict=sd, device=DEV, **kw)


def grad_summary(t):
    f = t.detach().float().flatten().cpu()
    n = min(256, f.numel())
    idx = (torch.arange(n, dtype=torch.long) * (f.numel() - 1)) // max(n - 1, 1)
    return torch.cat([f.norm()[None], f[idx]])


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


# ------------------------------------------------------------------ A5 on the device

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "a5_*.npz"))))
def test_prepare_inputs_matches_reference(path):
    g = np.load(path)
    Timg = int(g["rows_per_image"])
    left = bool(int(g["left"]))
    max_len = int(g["max_length"]) if int(g["max_length"]) >= 0 else None
    cfg = tiny_cfg(hidden_size=32, intermediate_size=64, num_image_tokens=Timg, tokenizer_model_max_length=max_len,
                   tokenizer_padding_side="left" if left else "right")
    sd = init_state_dict(cfg, seed=11, dtype=torch.bfloat16)
    model = hip_model(cfg, sd)
    if max_len is None:
        model.config.tokenizer_model_max_length = None
    N = int(g["num_images"])
    images = torch.randn(N, 3, 56, 56, generator=torch.Generator().manual_seed(5))
    ids, lab, msk = T(g["input_ids"]).to(DEV), T(g["labels"]).to(DEV), T(g["attention_mask"]).to(DEV)
    # a5_wrap_*: the wrapper's None handling (reference metamorph_arch.py:245-256, 400-412)
    if "labels_given" in g.files and not int(g["labels_given"]):
        lab = None
    if "mask_given" in g.files and not int(g["mask_given"]):
        msk = None
    pos_in = T(g["position_ids"]).to(DEV) if "position_ids" in g.files else None
    with torch.no_grad():
        proj, feat = model.encode_images(images.to(DEV))
        if "error" in g.files:                                # no tokenizer_model_max_length + an image: TypeError in the reference (:324)
            with pytest.raises(TypeError):
                model.prepare_inputs_labels_for_multimodal(ids, pos_in, msk, None, lab, images.to(DEV))
            return
        out = model.prepare_inputs_labels_for_multimodal(ids, pos_in, msk, None, lab, images.to(DEV))
    none_ids, pos_ids, att, _, emb, new_lab, img_pos, tgt = out
    assert none_ids is None
    if "out_position_ids" in g.files:                         # given position_ids come back REPLACED by arange over each sample's rows
        assert torch.equal(pos_ids.cpu(), T(g["out_position_ids"])) and pos_ids.dtype == torch.int64
    else:
        assert pos_ids is None
    if "out_labels" in g.files:
        assert torch.equal(new_lab.cpu(), T(g["out_labels"]))
    else:
        assert new_lab is None
    if "out_attention_mask" in g.files:
        assert torch.equal(att.cpu(), T(g["out_attention_mask"])) and att.dtype == torch.bool
    else:
        assert att is None
    assert torch.equal(img_pos.cpu(), T(g["out_image_positions"]))
    keep = g["out_target_keep"].tolist()
    assert tgt.shape[0] == len(keep)
    if keep:
        assert torch.equal(tgt.cpu(), feat.cpu()[keep])
    # every spliced row is a bit-exact copy of the row the reference took
    W = model.get_model().embed_tokens.weight.data
    flat = proj.reshape(-1, proj.shape[-1])
    src = g["out_src"]
    for b in range(src.shape[0]):
        for l in range(src.shape[1]):
            s = int(src[b, l])
            exp = W[s] if s >= 0 else (torch.zeros_like(W[0]) if s == -1 else flat[-2 - s])
            assert torch.equal(emb[b, l], exp), (b, l, s)


# ------------------------------------------------------------------ end to end vs golden + oracle

E2E = sorted(glob.glob(os.path.join(GOLDEN, "e2e_*_bf16.npz")))


@pytest.mark.parametrize("path", E2E)
def test_e2e_forward_backward(path):
    g = np.load(path)
    g32 = np.load(path.replace("_bf16", "_f32"))
    cfg = tiny_cfg(num_image_tokens=int(g["rows_per_image"]), use_vision_ar=bool(int(g["use_vision_ar"])),
                   normalize_vision=bool(int(g["normalize_vision"])), apply_softmax=bool(int(g["apply_softmax"])),
                   tokenizer_padding_side="left" if int(g["left"]) else "right", mm_projector_type=str(g["mm_projector_type"]),
                   image_token_reduction=str(g["image_token_reduction"]),
                   **({"vision_head_type": str(g["vision_head_type"])} if "vision_head_type" in g else {}),
                **({"vision_coef": float(g["vision_coef"])} if "vision_coef" in g else {}),
                **(json.loads(str(g["cfg_json"])) if "cfg_json" in g else {}))
    sd = init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16)
    model = hip_model(cfg, sd, vision_head=cfg.vision_head_type)
    model.train()
    ids, lab, msk = T(g["input_ids"]).to(DEV), T(g["labels"]).to(DEV), T(g["attention_mask"]).to(DEV)
    images = T(g["images"]).to(DEV)
    out = model(input_ids=ids, attention_mask=msk, labels=lab, images=images.bfloat16())
    ref_loss, truth = float(g["loss"]), float(g32["loss"])
    got = float(out.loss.detach())
    print(f"\n[{os.path.basename(path)}] loss hip={got:.6f} ref_bf16={ref_loss:.6f} ref_fp32={truth:.6f} "
          f"lang={model.loss_language:.6f}/{float(g['loss_language']):.6f} img={model.loss_image_ar:.6f}/{float(g['loss_image_ar']):.6f}")
    if np.isnan(ref_loss):
        assert np.isnan(got)
    else:
        assert abs(got - ref_loss) <= 1e-3 * abs(ref_loss), (got, ref_loss)
        assert abs(got - truth) <= 1e-3 * abs(truth), (got, truth, ref_loss)
    assert abs(model.loss_language - float(g["loss_language"])) <= 1e-3 * abs(float(g["loss_language"]))
    if np.isnan(float(g["loss_image_ar"])):
        assert np.isnan(model.loss_image_ar)          # no answer-side image rows -> NaN
    else:
        # image-AR head (cosine / mean-abs / soft-CE): as close to the fp32 truth as the reference's own bf16 run (x2), floor 1e-3
        li, li_ref, li_true = model.loss_image_ar, float(g["loss_image_ar"]), float(g32["loss_image_ar"])
        assert abs(li - li_true) <= max(1.5 * abs(li_ref - li_true), 1e-3 * max(1.0, abs(li_true))), (li, li_ref, li_true)
    # hidden states: as close to fp32 truth as the reference's own bf16 run (x2) -- valid rows only
    valid = T(np.asarray(out.hidden_states.shape[:2]))  # noqa
    hs = out.hidden_states.float().cpu()
    mask = torch.zeros(hs.shape[:2], dtype=torch.bool)
    L = hs.shape[1]
    # spliced attention mask from the oracle bookkeeping
    o32 = oracle_forward(init_state_dict(cfg, seed=int(g["seed"])), cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]),
                         T(g["images"]), return_logits=False)
    mask = o32["attention_mask"]
    assert torch.equal(mask, torch.zeros_like(mask)) is False and (not int(g["left"]) or bool(mask[:, -1].all()))   # left: valid rows end at L
    e_hip = rel(hs[mask], T(g32["hidden"])[mask])
    e_ref = rel(T(g["hidden"])[mask], T(g32["hidden"])[mask])
    print(f"   hidden rel err vs fp32: hip={e_hip:.4e} reference-bf16={e_ref:.4e}")
    assert e_hip <= max(1.5 * e_ref, 1.3e-2)                       # measured 8.0e-3 .. 8.4e-3 (reference-bf16: 8.4e-3 .. 8.9e-3)
    if np.isnan(ref_loss):
        return
    out.loss.backward()
    worst = 0.0
    n = 0
    for k in g32.files:
        if not k.startswith("grad::"):
            continue
        name = k[6:]
        p = dict(model.named_parameters())[name]
        assert p.grad is not None, name
        got_g = grad_summary(p.grad)
        e_h = rel(got_g[1:], T(g32[k])[1:])
        e_r = rel(T(g[k])[1:], T(g32[k])[1:]) if k in g.files else 0.0
        nerr = abs(float(got_g[0]) - float(g32[k][0])) / max(float(g32[k][0]), 1e-12)
        if e_h > 2e-2:
            print(f"      {name}: rel err vs fp32 hip={e_h:.3e} reference-bf16={e_r:.3e} norm err={nerr:.3e}")
        # tensors whose gradient is near-noise in the reference's own bf16 run (embed_tokens rows, the l1 / soft-CE heads) are held to
        # 3 x that run's distance from fp32; everything else to 3.3e-2 = 1.5 x the worst measured (2.2e-2)
        ok = e_h <= max(3.0 * e_r, 3.3e-2)
        nz = int((T(g32[k])[1:] != 0).sum())
        if not ok and nz <= 4:
            # a sparse gradient (embed_tokens: ~40 used rows of 128 258) of which the 256-entry sample holds ONE element, 40 x smaller
            # than the tensor's typical entry: its relative error is noise (the reference's own bf16 run is 3-11 % off on it).  Judge
            # the element on the scale of the tensor's non-zero entries and the tensor by its norm instead.
            gf = p.grad.detach().float()
            rms = float(gf[gf != 0].pow(2).mean().sqrt())
            abs_err = float((got_g[1:] - T(g32[k])[1:]).abs().max())
            print(f"      {name}: sparse sample ({nz} non-zero): |err| / rms of the non-zero entries = {abs_err / rms:.3e}, norm err {nerr:.3e}")
            ok = abs_err <= 3.3e-2 * rms and nerr <= 1e-2
        else:
            worst = max(worst, e_h)
        assert ok, (name, e_h, e_r)
        assert nerr <= max(3e-2, 3 * abs(float(g[k][0]) - float(g32[k][0])) / max(float(g32[k][0]), 1e-12)), (name, nerr)
        n += 1
    print(f"   {n} gradient tensors checked, worst rel err vs fp32 truth {worst:.3e}")
    assert n >= 20


@pytest.mark.parametrize("kind", ["image_embeds", "pretraining_tp2"])
def test_image_embeds_and_pretraining_tp_match_reference_recorded(kind):
    """`forward(image_embeds=...)` -> `encode_imagesembed` (reference metamorph_arch.py:166-173, metamorph_llama.py:603-660: tower
    skipped, the given [N, T, 1152] features go through mm_projector and are the regression targets) and `pretraining_tp = 2`
    (metamorph_llama.py:393-396) against runs of the reference itself (tests/golden/r3_*.npz, oracle/gen_golden.py r3)."""
    g, g32 = np.load(os.path.join(GOLDEN, f"r3_{kind}_bf16.npz")), np.load(os.path.join(GOLDEN, f"r3_{kind}_f32.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    model.config.pretraining_tp = int(g["pretraining_tp"])
    model.train()
    ids, lab, msk = T(g["input_ids"]).to(DEV), T(g["labels"]).to(DEV), T(g["attention_mask"]).to(DEV)
    kw = dict(image_embeds=T(g["image_embeds"]).to(DEV).bfloat16()) if kind == "image_embeds" else dict(images=T(g["images"]).to(DEV).bfloat16())
    out = model(input_ids=ids, attention_mask=msk, labels=lab, **kw)
    got, ref_loss, truth = float(out.loss.detach()), float(g["loss"]), float(g32["loss"])
    print(f"\n   [{kind}] loss hip={got:.6f} ref_bf16={ref_loss:.6f} ref_fp32={truth:.6f} img={model.loss_image_ar:.6f}/{float(g32['loss_image_ar']):.6f}")
    assert abs(got - ref_loss) <= 1e-3 * abs(ref_loss) and abs(got - truth) <= 1e-3 * abs(truth)
    assert abs(model.loss_language - float(g32["loss_language"])) <= 1e-3 * abs(truth)
    assert abs(model.loss_image_ar - float(g32["loss_image_ar"])) <= 1e-3
    plan = model.prepare_inputs_labels_for_multimodal(ids, None, msk, None, lab, kw.get("images"), image_embeds=kw.get("image_embeds"))
    mask = plan[2].bool().cpu()
    if kind == "image_embeds":                                  # the targets ARE the given features (answer images only), bit for bit
        assert torch.equal(plan[7].float().cpu(), T(g["target_features"]))
        proj, tgt = model.encode_imagesembed(kw["image_embeds"])
        assert torch.equal(tgt, kw["image_embeds"]) and proj.shape == (tgt.shape[0], 4, cfg.hidden_size)
    e_hip, e_ref = rel(out.hidden_states.float().cpu()[mask], T(g32["hidden"])[mask]), rel(T(g["hidden"])[mask], T(g32["hidden"])[mask])
    assert e_hip <= max(1.5 * e_ref, 1.3e-2), (e_hip, e_ref)
    out.loss.backward()
    params, n = dict(model.named_parameters()), 0
    for k in g32.files:
        if k.startswith("grad::"):
            gs = grad_summary(params[k[6:]].grad)
            e_h, e_r = rel(gs[1:], T(g32[k])[1:]), rel(T(g[k])[1:], T(g32[k])[1:])
            assert e_h <= max(3.0 * e_r, 3.3e-2), (k, e_h, e_r)
            n += 1
    assert n >= 25
    model.eval()
    with torch.no_grad():                                       # the full fp32 logits (sliced or not: the same numbers)
        lo = model(input_ids=ids, attention_mask=msk, labels=None, **kw).logits[:, :, ::997].cpu()
    e_hip, e_ref = rel(lo[mask], T(g32["logits_sub"])[mask]), rel(T(g["logits_sub"])[mask], T(g32["logits_sub"])[mask])
    assert e_hip <= max(1.5 * e_ref, 1.4e-2), (e_hip, e_ref)
    if kind == "pretraining_tp2":
        model.config.pretraining_tp = 4                         # 128258 % 4 != 0: the reference would silently drop the last two rows
        with pytest.raises(NotImplementedError):
            model(input_ids=ids, attention_mask=msk, labels=None, **kw)


def test_generate_with_image_embeds_equals_generate_with_images():
    """`generate(image_embeds=tower(images))` (reference metamorph_llama.py:672-700) walks the same loop as `generate(images=...)`."""
    from oracle.ref_model import decode_fixture_state_dict
    g = np.load(os.path.join(GOLDEN, "n1_decode_image_prompt.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, decode_fixture_state_dict(g, cfg, torch.bfloat16)).eval()
    images = T(g["images"]).to(DEV).bfloat16()
    with torch.no_grad():
        feats = model.get_model().get_vision_tower()(images)
    a, ea = model.generate(inputs=T(g["input_ids"]).to(DEV), images=images, output_image=True, max_new_tokens=int(g["max_new_tokens"]))
    b, eb = model.generate(inputs=T(g["input_ids"]).to(DEV), image_embeds=feats, output_image=True, max_new_tokens=int(g["max_new_tokens"]))
    assert a[0].tolist() == b[0].tolist() == g["tokens"].tolist() and torch.equal(ea, eb)


def test_mean_abs_head_with_unequal_row_counts_follows_reference():
    """`mse_loss_fn` (reference metamorph_llama.py:211-219) zips target and prediction rows and divides by len(target): with R != Rt
    it uses the first min(R, Rt) pairs (oracle pinned to the reference by tests/golden/ops_r3.npz).  Driven through llm_forward with a
    target tensor shorter / longer than the prediction rows; value and gradient against the oracle on the device model's own rows."""
    from oracle import ref_ops as R
    gg = np.load(os.path.join(GOLDEN, "e2e_generation_only_T4_ar1_l1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4, normalize_vision=False)
    sd16 = init_state_dict(cfg, seed=int(gg["seed"]), dtype=torch.bfloat16)
    model = hip_model(cfg, sd16)
    model.train()
    ids, lab, msk = T(gg["input_ids"]).to(DEV), T(gg["labels"]).to(DEV), T(gg["attention_mask"]).to(DEV)
    images = T(gg["images"]).to(DEV).bfloat16()
    for rows in (6, 8, 11):                                     # R = 8 prediction rows; fewer, equal, more target rows
        (_, _, amask, _, emb, nlab, ipos, tgt) = model.prepare_inputs_labels_for_multimodal(ids, None, msk, None, lab, images)
        flat = tgt.reshape(-1, tgt.shape[-1])
        assert flat.shape[0] == 8
        t_in = torch.cat([flat, flat[:3]], 0)[:rows].contiguous().view(1, rows, -1)
        model.zero_grad(set_to_none=True)
        out = model.llm_forward(inputs_embeds=emb, attention_mask=amask, labels=nlab, image_positions=ipos, image_features=t_in)
        out.loss.backward()
        hid = out.hidden_states.detach()
        sel = ipos[:, 1:].bool()
        pred_in = hid[:, :-1][sel].float().cpu().requires_grad_(True)
        w = {k: v.float() for k, v in sd16.items() if k.startswith("vision_head.")}
        from oracle.ref_model import vision_head
        want = R.mean_abs_loss(t_in.view(rows, -1).float().cpu(), vision_head(w, cfg, pred_in))
        assert abs(model.loss_image_ar - float(want)) <= 2e-3 * abs(float(want)), (rows, model.loss_image_ar, float(want))
        for k, v in w.items():
            v.requires_grad_(True)
        want2 = R.mean_abs_loss(t_in.view(rows, -1).float().cpu(), vision_head(w, cfg, pred_in.detach()))
        want2.backward()
        e = rel(dict(model.named_parameters())["vision_head.2.bias"].grad, w["vision_head.2.bias"].grad)
        assert e <= 6e-2, (rows, e)                               # sign gradients scaled by min(R, Rt) / (Rt C): scale errors would be >= 25 %


def test_logits_eval_mode():
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    g32 = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_f32.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    model.eval()
    with torch.no_grad():
        out = model(input_ids=T(g["input_ids"]).to(DEV), attention_mask=T(g["attention_mask"]).to(DEV),
                    labels=None, images=T(g["images"]).to(DEV).bfloat16())
    assert out.loss is None and out.logits.dtype == torch.float32 and out.logits.shape[-1] == 128258
    # padded positions are outside the contract (the loss ignores them; this build zeroes their attention output)
    o32 = oracle_forward(init_state_dict(cfg, seed=int(g["seed"])), cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]),
                         T(g["images"]), return_logits=False)
    mask = o32["attention_mask"]
    sub = out.logits[:, :, ::997].cpu()
    e_hip, e_ref = rel(sub[mask], T(g32["logits_sub"])[mask]), rel(T(g["logits_sub"])[mask], T(g32["logits_sub"])[mask])
    print(f"\n   logits rel err vs fp32: hip={e_hip:.4e} reference-bf16={e_ref:.4e}")
    assert e_hip <= max(1.5 * e_ref, 1.4e-2)                       # measured 9.0e-3 (reference-bf16 9.4e-3)


def test_explicit_position_ids_are_checked_not_ignored():
    """The reference hands position_ids through to HF; this build rotates by the row index, which is the same attention for any
    positions that advance by one over a sample's valid rows -- and refuses anything else instead of ignoring it."""
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    model.eval()
    ids, msk, img = T(g["input_ids"]).to(DEV), T(g["attention_mask"]).to(DEV), T(g["images"]).to(DEV).bfloat16()
    with torch.no_grad():
        # (forward(input_ids=...) rebuilds position_ids itself, like the reference; the check guards callers that bring inputs_embeds)
        _, _, amask, _, emb, _, _, _ = model.prepare_inputs_labels_for_multimodal(ids, None, msk, None, None, img, None, None)
        B, L = emb.shape[:2]
        pos = torch.arange(L, device=DEV)[None].expand(B, -1).contiguous()
        base = model(inputs_embeds=emb, attention_mask=amask, labels=None)
        same = model(inputs_embeds=emb, attention_mask=amask, position_ids=pos + 7, labels=None)     # a constant offset is fine
        assert torch.equal(base.logits, same.logits)
        bad = pos.clone()
        bad[:, 5:] += 3                                              # a gap inside every sample
        with pytest.raises(NotImplementedError):
            model(inputs_embeds=emb, attention_mask=amask, position_ids=bad, labels=None)


def test_grad_accumulation_doubles():
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    model.train()
    args = dict(input_ids=T(g["input_ids"]).to(DEV), attention_mask=T(g["attention_mask"]).to(DEV),
                labels=T(g["labels"]).to(DEV), images=T(g["images"]).to(DEV).bfloat16())
    model(**args).loss.backward()
    first = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
    model(**args).loss.backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert rel(p.grad, 2 * first[n]) < 2e-2, n
    model.zero_grad(set_to_none=True)
    model(**args).loss.backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert rel(p.grad, first[n]) < 1e-2, n


def test_paired_weight_gradient_launch_matches_separate_launches(monkeypatch):
    """DecoderLayerFn.backward sends the down_proj and qkv weight gradients out as one launch when that saves a wave of
    workgroups (LLaMA-3-8B); force that path on a tiny model (128 token rows) and compare every gradient with the unpaired run."""
    import metamorph_amd.functional as F
    cfg = tiny_cfg(num_image_tokens=4)
    sd = init_state_dict(cfg, seed=11, dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 127000, (2, 64), generator=gen).to(DEV)
    args = dict(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids.clone())

    def grads(paired):
        calls = []
        orig = F.ops.gemm_pair
        monkeypatch.setitem(F.VARIANTS, "dw_pair", paired)
        monkeypatch.setattr(F, "_pair_saves_a_wave", lambda r0, c0, r1, c1, k: ((k + 7) // 8 * 8) % 128 == 0)
        monkeypatch.setattr(F.ops, "gemm_pair", lambda *a: (calls.append(1), orig(*a))[1])
        model = hip_model(cfg, sd)
        model.train()
        for _ in range(2):                                   # second pass: the accumulate flavour of both problems
            model(**args).loss.backward()
        monkeypatch.setattr(F.ops, "gemm_pair", orig)
        return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}, len(calls)

    ref, n_ref = grads(False)
    got, n_got = grads(True)
    assert n_ref == 0 and n_got == 2 * cfg.num_hidden_layers, (n_ref, n_got)
    assert got.keys() == ref.keys()
    for n in ref:
        assert rel(got[n], ref[n]) < 2e-3, (n, rel(got[n], ref[n]))


def test_transposed_weight_cache_under_accumulation(monkeypatch):
    """functional.transposed_weight with the cache on (the Trainer turns it on when gradient_accumulation_steps > 1): the second
    micro-batch's backward re-uses the transposed decoder weights of the first (4 transposes per layer less), gradients carry the same
    bits as without the cache, and an in-place write to the parameters through torch refreshes the copies WITHOUT anybody bumping the
    parameter generation (round 4 needed the bump: a stale copy was one forgotten call away; the optimizers, which write behind torch's
    version counters, still bump)."""
    import metamorph_amd.functional as F
    cfg = tiny_cfg(num_image_tokens=4)
    sd = init_state_dict(cfg, seed=11, dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 127000, (2, 64), generator=gen).to(DEV)
    args = dict(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids.clone())

    def run(cache):
        monkeypatch.setitem(F.VARIANTS, "wt_cache", cache)
        F.drop_transposed_weights()
        calls = []
        orig = F.ops.transpose
        monkeypatch.setattr(F.ops, "transpose", lambda x, **kw: (calls.append(tuple(x.shape)), orig(x, **kw))[1])
        model = hip_model(cfg, sd)
        model.train()
        per_pass = []
        for _ in range(2):                                   # two micro-batches of one accumulation window
            n0 = len(calls)
            model(**args).loss.backward()
            per_pass.append(len(calls) - n0)
        g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        with torch.no_grad():                                # a torch-side write (no generation bump): the version counters invalidate the copies
            for p in model.model.layers.parameters():
                p.mul_(0.5)
        model.zero_grad(set_to_none=True)
        model(**args).loss.backward()
        g2 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        monkeypatch.setattr(F.ops, "transpose", orig)
        return per_pass, g1, g2

    p_off, a1, a2 = run(False)
    p_on, b1, b2 = run(True)
    F.drop_transposed_weights()
    print(f"\n   transposes per backward call: cache off {p_off}, cache on {p_on}")
    assert p_off[0] == p_off[1] and p_on[0] <= p_off[0], (p_on, p_off)
    assert p_on[1] <= p_off[1] - 4 * cfg.num_hidden_layers, (p_on, p_off)     # down, gate|up, o, q|k|v per layer re-used
    for n in a1:
        assert torch.equal(a1[n], b1[n]), n
        assert torch.equal(a2[n], b2[n]), n                  # stale copies would show here


def test_stage1_freeze_policy():
    """Only mm_projector (+ embed_tokens) trainable, as in the reference's stage 1 (train.py:1515-1519)."""
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    for n, p in model.named_parameters():
        p.requires_grad_("mm_projector" in n or "embed_tokens" in n)
    model.train()
    out = model(input_ids=T(g["input_ids"]).to(DEV), attention_mask=T(g["attention_mask"]).to(DEV),
                labels=T(g["labels"]).to(DEV), images=T(g["images"]).to(DEV).bfloat16())
    out.loss.backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad.float()).all() and float(p.grad.float().abs().max()) > 0, n
        else:
            assert p.grad is None, n


# ------------------------------------------------------------------ cached decode
def _decode_model(**kw):
    cfg = tiny_cfg(num_key_value_heads=1, **kw)
    sd = init_state_dict(cfg, seed=5)
    return cfg, hip_model(cfg, sd).eval()


def test_cached_decode_matches_full_forward():
    """Prefill 21 rows, then feed rows 21..39 one at a time through the decode-shape kernels (GEMV + KV-cache attention):
    every step's logits must agree with the logits of ONE full forward over all 40 rows at that position."""
    from metamorph_amd import functional as F
    cfg, model = _decode_model()
    h = cfg.hidden_size
    L, L0 = 40, 21
    g = torch.Generator().manual_seed(3)
    emb = (torch.randn(1, L, h, generator=g) * 0.5).bfloat16().to(DEV)
    with torch.no_grad():
        full = model.llm_forward(inputs_embeds=emb, return_dict=True).logits[0]            # [L, V] fp32
        _, meta = model._decode_meta(L0)
        cos, sin = model.model.rope_tables(L + 2, DEV)
        meta.cos, meta.sin = cos, sin
        cache = F.KVCache(cfg.num_hidden_layers, L + 2, meta.Hkv * meta.d, DEV)
        x = F.decoder_prefill(emb[0, :L0].contiguous(), model.model.layers, meta, cache)[-1:].contiguous()
        scale = float(full.abs().max())
        for t in range(L0 - 1, L):
            logits, _, _ = model._head_row(x, False)
            err = float((logits[0] - full[t]).abs().max())
            assert err <= 2e-2 * scale, f"position {t}: cached logits differ by {err} (scale {scale})"
            assert int(logits[0].argmax()) == int(full[t].argmax()) or float(full[t].topk(2).values.diff().abs()) < 2e-2 * scale
            if t + 1 < L:
                x = F.decoder_decode_row(emb[0, t + 1:t + 2].contiguous(), model.model.layers, meta, cache, cos, sin)
        assert cache.length == L


def test_prompt_pass_split_k_equals_the_plain_projections():
    """The prompt pass of a cached decode sends q|k|v, o and down through the split-K GEMM (functional.VARIANTS["prefill_splitk"]; h = 1024,
    300 rows: two K slices): hidden rows and the cached K / V rows against the same pass on the plain kernels -- one bf16 step (another fp32
    summation order), and the continuation of the decode reads the same argmax."""
    from metamorph_amd import functional as F
    cfg = tiny_cfg(hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=2, num_hidden_layers=3)
    model = hip_model(cfg, init_state_dict(cfg, seed=23, dtype=torch.bfloat16)).eval()
    L0 = 300
    g = torch.Generator().manual_seed(8)
    emb = (torch.randn(L0, cfg.hidden_size, generator=g) * 0.5).bfloat16().to(DEV)
    outs = {}
    with torch.no_grad():
        for on in (True, False):
            old = F.set_variant("prefill_splitk", on)
            try:
                _, meta = model._decode_meta(L0)
                cos, sin = model.model.rope_tables(L0 + 4, DEV)
                meta.cos, meta.sin = cos, sin
                cache = F.KVCache(cfg.num_hidden_layers, L0 + 4, meta.Hkv * meta.d, DEV)
                x = F.decoder_prefill(emb.clone(), model.model.layers, meta, cache)
                assert meta.prompt_pass is on
                outs[on] = (x.float().clone(), cache.k[:, 0, :L0].float().clone(), cache.v[:, 0, :L0].float().clone(), model._head_row(x[-1:].contiguous(), False)[0].clone())
            finally:
                F.set_variant("prefill_splitk", old)
    for a, b, what in zip(outs[True][:3], outs[False][:3], ("hidden rows", "cached k", "cached v")):
        assert rel(a, b) < 6e-3, (what, rel(a, b))
        assert not torch.equal(a, b) or what != "hidden rows"      # the split form really ran (another summation order)
    assert int(outs[True][3].argmax()) == int(outs[False][3].argmax())


def test_prompt_pass_fused_reduce_launches_give_the_same_bits():
    """The prompt pass of one sequence with RoPE + the cache rows and both RMSNorms in the reduce launches of the split projections, attention
    reading K / V from the cache (functional.VARIANTS["prefill_fused"]) against the same pass with them as launches of their own: hidden rows
    and cached K / V rows bit for bit (h = 1024; 300, 77 and 40 rows: split widths); the rows of the cache behind the prompt stay untouched."""
    from metamorph_amd import functional as F
    cfg = tiny_cfg(hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=2, num_hidden_layers=3)
    model = hip_model(cfg, init_state_dict(cfg, seed=29, dtype=torch.bfloat16)).eval()
    for L0 in (300, 77, 40):                                       # (40 rows: gate|up takes the split-K GEMM too)
        g = torch.Generator().manual_seed(L0)
        emb = (torch.randn(L0, cfg.hidden_size, generator=g) * 0.5).bfloat16().to(DEV)
        outs = {}
        with torch.no_grad():
            for on in (True, False):
                old = F.set_variant("prefill_fused", on)
                try:
                    _, meta = model._decode_meta(L0)
                    cos, sin = model.model.rope_tables(L0 + 4, DEV)
                    meta.cos, meta.sin = cos, sin
                    cache = F.KVCache(cfg.num_hidden_layers, L0 + 4, meta.Hkv * meta.d, DEV)
                    cache.k.fill_(7.0); cache.v.fill_(-7.0)
                    x = F.decoder_prefill(emb.clone(), model.model.layers, meta, cache)
                    assert meta.prompt_pass and cache.lengths == [L0]
                    outs[on] = (x.clone(), cache.k.clone(), cache.v.clone())
                finally:
                    F.set_variant("prefill_fused", old)
        for a, b, what in zip(outs[True], outs[False], ("hidden rows", "cache k", "cache v")):
            assert torch.equal(a, b), (L0, what, float((a.float() - b.float()).abs().max()))
        assert float(outs[True][1][:, 0, L0:].min()) == 7.0 and float(outs[True][2][:, 0, L0:].max()) == -7.0


def test_cached_image_mode_head_matches_reference_loop_step():
    """Image mode: vision_head -> L2 norm -> mm_projector on the last row; cached head == llm_forward(decoding=True)."""
    cfg, model = _decode_model()
    g = torch.Generator().manual_seed(4)
    emb = (torch.randn(1, 17, cfg.hidden_size, generator=g) * 0.5).bfloat16().to(DEV)
    from metamorph_amd import functional as F
    with torch.no_grad():
        out = model.llm_forward(inputs_embeds=emb, return_dict=True, decoding=True)
        _, meta = model._decode_meta(17)
        cos, sin = model.model.rope_tables(32, DEV)
        meta.cos, meta.sin = cos, sin
        cache = F.KVCache(cfg.num_hidden_layers, 32, meta.Hkv * meta.d, DEV)
        x = F.decoder_prefill(emb[0].contiguous(), model.model.layers, meta, cache)[-1:].contiguous()
        logits, fed_back, pred_z = model._head_row(x, True)
    assert rel(pred_z, out.loss) < 2e-2 and rel(fed_back, out.hidden_states[:, -1]) < 2e-2
    assert rel(logits[0], out.logits[0, -1]) < 2e-2


def test_greedy_decode_cached_equals_reprefill():
    """The reference's loop (re-run the prefix every step) and the KV-cache loop walk the same state machine."""
    cfg, model = _decode_model()
    g = torch.Generator().manual_seed(6)
    emb = (torch.randn(1, 12, cfg.hidden_size, generator=g) * 0.5).bfloat16().to(DEV)
    a = model.greedy_decode(None, None, emb, max_new_tokens=8, use_cache=True)[0]
    b = model.greedy_decode(None, None, emb, max_new_tokens=8, use_cache=False)[0]
    assert a.shape == b.shape and a.numel() >= 1
    same = (a == b).nonzero().numel()
    assert same >= a.numel() - 1 and int(a[0]) == int(b[0])     # bf16 near-ties may flip a late token, never the first
    # the hipGraph replay of the decode step and eager launches are the same kernels on the same data: identical tokens
    from metamorph_amd import functional as F_
    old = F_.set_variant("decode_graph", False)
    try:
        c = model.greedy_decode(None, None, emb, max_new_tokens=8, use_cache=True)[0]
    finally:
        F_.set_variant("decode_graph", old)
    assert torch.equal(a, c)


@pytest.mark.parametrize("use_cache", [True, False])
@pytest.mark.parametrize("name", ["text", "image_prompt", "image_prompt_rope31"])
def test_greedy_decode_matches_reference_recorded_loop(name, use_cache):
    """Row N1 pinned to the reference: tests/golden/n1_decode_*.npz hold what the reference's OWN `generate` -> `greedy_decode`
    (metamorph_llama.py:665-717, 502-597) emitted on these weights -- token mode -> <image_start> -> four continuous image tokens
    fed back through vision_head / mm_projector -> <image_end> -> text -> <|eot_id|> (decision margins > 6 logits, so bf16 cannot
    flip them).  Both HIP loops (KV cache + hipGraph, and the reference-style re-prefill) must emit the same ids and the same
    `pred_z` rows (as close to the fp32 run as the reference's bf16 run, x2)."""
    from oracle.ref_model import decode_fixture_state_dict
    g = np.load(os.path.join(GOLDEN, f"n1_decode_{name}.npz"))
    cfg = tiny_cfg(num_image_tokens=4, **(json.loads(str(g["cfg_json"])) if "cfg_json" in g else {}))     # *_rope31: LLaMA-3.1 RoPE
    model = hip_model(cfg, decode_fixture_state_dict(g, cfg, torch.bfloat16)).eval()
    assert model.model.rope.rope_type == ((cfg.rope_scaling or {}).get("rope_type", "default"))
    images = T(g["images"]).to(DEV).bfloat16() if g["images"].size else None
    out, emb = model.generate(inputs=T(g["input_ids"]).to(DEV), images=images, output_image=True,
                              max_new_tokens=int(g["max_new_tokens"]), use_cache=use_cache)
    assert out[0].dtype == torch.int32 and out[0].tolist() == g["tokens"].tolist(), (out[0].tolist(), g["tokens"].tolist())
    assert emb.shape == tuple(g["pred_z"].shape)
    e_hip, e_ref = rel(emb, T(g["pred_z"])), rel(T(g["pred_z_bf16"]), T(g["pred_z"]))
    print(f"\n   [{name} use_cache={use_cache}] pred_z rel err vs reference fp32: hip={e_hip:.3e} reference-bf16={e_ref:.3e}")
    assert e_hip <= max(1.5 * e_ref, 1.2e-2)                       # measured 7.6e-3 .. 8.3e-3 (reference-bf16 7.9e-3 .. 8.2e-3)
    for r in range(emb.shape[0]):                                   # every image-mode step, not only the average
        assert rel(emb[r], T(g["pred_z"])[r]) <= max(3.0 * e_ref, 2e-2), r
    # `max_new_tokens` ends the loop after max_new + 1 iterations, image-mode iterations included (reference :587-590): 2 stops inside the
    # image (one id, two pred_z rows), 6 after <image_end> and one text token; `output_image=False` returns the id list alone
    for mn in (2, 6):
        o_m, e_m = model.generate(inputs=T(g["input_ids"]).to(DEV), images=images, output_image=True, max_new_tokens=mn, use_cache=use_cache)
        assert o_m[0].tolist() == g[f"tokens_max{mn}"].tolist() and e_m.shape[0] == int(g[f"n_pred_z_max{mn}"]), (mn, o_m[0].tolist(), e_m.shape)
        assert e_m.shape[0] == 0 or torch.equal(e_m, emb[:e_m.shape[0]])
    only = model.generate(inputs=T(g["input_ids"]).to(DEV), images=images, max_new_tokens=2, use_cache=use_cache)
    assert isinstance(only, list) and len(only) == 1 and only[0].tolist() == g["tokens_max2"].tolist()


@pytest.mark.parametrize("name", ["text", "image_prompt", "text_rope31"])
def test_hf_generate_matches_reference_recorded(name):
    """`generate(use_customize_greedy=False, ...)` (reference metamorph_llama.py:711-717: transformers' GenerationMixin driving forward
    with a KV cache): greedy search and top-p sampling must emit the ids the REFERENCE's own HF-generate run emitted on these weights
    (tests/golden/hfgen_*.npz, oracle/gen_golden.py hfgen; decision margins > 6 logits; at T = 0.7 the planned token holds > 0.99 of
    the mass, so the nucleus is one token and sampling is seed-independent).  Here the cache is a HipKVCache: prompt pass on the
    training-path kernels, then one row per step through the decode kernels (hipGraph replay); per-step logits of the active
    tokens as close to the reference's fp32 run as its own bf16 run (x 1.5)."""
    from oracle.ref_model import decode_fixture_state_dict
    from metamorph_amd.model.language_model.metamorph_llama import HipKVCache
    g = np.load(os.path.join(GOLDEN, f"hfgen_{name}.npz"))
    cfg = tiny_cfg(num_image_tokens=4, **(json.loads(str(g["cfg_json"])) if "cfg_json" in g else {}))
    model = hip_model(cfg, decode_fixture_state_dict(g, cfg, torch.bfloat16)).eval()
    images = T(g["images"]).to(DEV).bfloat16() if g["images"].size else None
    ids = T(g["input_ids"]).to(DEV)
    kw = dict(inputs=ids, images=images, use_customize_greedy=False, max_new_tokens=int(g["max_new_tokens"]), eos_token_id=128009,
              pad_token_id=128001)
    out = model.generate(do_sample=False, output_scores=True, return_dict_in_generate=True, **kw)
    assert out.sequences[0].tolist() == g["tokens"].tolist(), (out.sequences[0].tolist(), g["tokens"].tolist())
    assert isinstance(out.past_key_values, HipKVCache) and out.past_key_values.get_seq_length() >= len(g["tokens"]) - 1
    act = g["active"].tolist()
    got = torch.stack([sc[0, act].float().cpu() for sc in out.scores])
    e_hip, e_ref = rel(got, T(g["active_logits"])), rel(T(g["active_logits_bf16"]), T(g["active_logits"]))
    print(f"\n   [hf generate {name}] active-token logits rel err vs reference fp32: hip={e_hip:.3e} reference-bf16={e_ref:.3e}")
    assert e_hip <= max(1.5 * e_ref, 1.5e-2)
    for seed in (0, 3):
        torch.manual_seed(seed)
        smp = model.generate(do_sample=True, temperature=0.7, top_p=0.9, **kw)
        assert smp[0].tolist() == g["sampled_tokens"].tolist()
    # beam search: two hypotheses returned, the second leaves the planned path, so HF re-orders the cache rows (HipKVCache.reorder_cache)
    bo = model.generate(num_beams=2, num_return_sequences=2, do_sample=False, return_dict_in_generate=True, output_scores=True, **kw)
    assert bo.sequences.tolist() == g["beam_sequences"].tolist(), (bo.sequences.tolist(), g["beam_sequences"].tolist())
    sc, ref32, ref16 = bo.sequences_scores.float().cpu(), T(g["beam_scores"]), T(g["beam_scores_bf16"])
    print(f"   [hf generate {name}] beam scores hip={sc.tolist()} reference fp32={ref32.tolist()} bf16={ref16.tolist()}")
    # measured on MI355X: 1.5e-3 (text) / 1.11e-2 (image prompt, second beam: a -2.32 sum of log-probabilities) where the reference's own bf16
    # run is 3.1e-3 / 4.2e-3 off its fp32 run; floor = 1.5 x the larger measurement
    assert float((sc - ref32).abs().max()) <= max(3.0 * float((ref16 - ref32).abs().max()), 1.7e-2)


def test_hf_generate_batch_of_left_padded_prompts_on_device():
    """Batched `generate(use_customize_greedy=False)` (reference metamorph_llama.py:711-717): two prompts of different lengths, the shorter
    LEFT-padded, with their attention mask -- on the device through the HIP kernels (round 3 ran this path only with the compute hooks
    replaced by the oracle on the CPU).  Every row must emit what the reference recorded for the batch (hfgen_text.npz; each row
    generates what it generates alone), the padding rows are never cached, the short prompt alone walks through the same per-step
    arithmetic (the batch is decoded in ONE pass per step since round 5 -- GEMV kernels at M = 2 rows instead of M = 1: same products per
    row; the scores are compared at accumulation-order accuracy and reported when bit-identical), right padding is refused."""
    from oracle.ref_model import decode_fixture_state_dict
    from metamorph_amd.model.language_model.metamorph_llama import HipKVCache
    g = np.load(os.path.join(GOLDEN, "hfgen_text.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, decode_fixture_state_dict(g, cfg, torch.bfloat16)).eval()
    ids, mask = T(g["batch_input_ids"]).to(DEV), T(g["batch_attention_mask"]).to(DEV)
    assert not bool(mask[1, 0]) and bool(mask[1, -1])             # row 1 is left-padded
    kw = dict(use_customize_greedy=False, do_sample=False, max_new_tokens=6, eos_token_id=128009, pad_token_id=128001, return_dict_in_generate=True)
    out = model.generate(inputs=ids, attention_mask=mask, output_scores=True, **kw)
    assert out.sequences.tolist() == g["batch_sequences"].tolist(), (out.sequences.tolist(), g["batch_sequences"].tolist())
    assert out.sequences[1].tolist() == g["short_alone_sequence"].tolist()
    assert isinstance(out.past_key_values, HipKVCache)
    st = out.past_key_values.states
    n_pad = int((~mask[1]).sum())
    assert [s_.pad for s_ in st] == [0, n_pad] and st[0].length == st[1].length + n_pad      # pad rows were never computed or cached
    alone = model.generate(inputs=ids[1:, n_pad:], output_scores=True, **kw)
    assert alone.sequences[0].tolist() == out.sequences[1].tolist()
    for step, (a, b) in enumerate(zip(alone.scores, out.scores)):
        if step == 0:
            # the prompt pass sends the batch's 2 x n rows through the MFMA GEMM of lm_head and the lone short prompt's <= 8 rows through the
            # GEMV (ops dispatch on the row count): same products, different fp32 summation order
            assert torch.allclose(a[0], b[1], rtol=1e-4, atol=2e-3), float((a[0] - b[1]).abs().max())
        else:
            assert torch.allclose(a[0], b[1], rtol=1e-4, atol=2e-3), (step, float((a[0] - b[1]).abs().max()))
    print(f"\n   batched vs alone, decoded steps bit-identical: {all(torch.equal(a[0], b[1]) for a, b in list(zip(alone.scores, out.scores))[1:])}")
    with pytest.raises(NotImplementedError):                      # right padding would put pad rows between the prompt and the generated tokens
        model.generate(inputs=ids.flip(1), attention_mask=mask.flip(1), use_customize_greedy=False, do_sample=False, max_new_tokens=2,
                       eos_token_id=128009, pad_token_id=128001)


def test_rope31_long_sample_real_llama31_constants_on_device():
    """The reference's forward + backward on ONE 4096-row sample under the REAL LLaMA-3.1 RoPE constants (rope_type "llama3", factor 8,
    context 8192 -- README.md:178,187's base model), recorded in tests/golden/r6_rope31_long_*.npz with attention sharp enough that
    positions matter: under the DEFAULT RoPE the same weights put the hidden rows 0.75 (relative) away, so a table that ignored
    rope_scaling cannot pass.  HIP bf16 must be as close to the reference's fp32 run as the reference's own bf16 run (x 1.5)."""
    from test_oracle_vs_golden import long_rope_state_dict
    g, g32 = (np.load(os.path.join(GOLDEN, f"r6_rope31_long_{t}.npz")) for t in ("bf16", "f32"))
    assert float(g32["default_rope_hidden_rel_far"]) > 0.5
    cfg = tiny_cfg(num_image_tokens=int(g["rows_per_image"]), **json.loads(str(g["cfg_json"])))
    model = hip_model(cfg, long_rope_state_dict(cfg, g, torch.bfloat16))
    assert model.model.rope.rope_type == "llama3"
    model.train()
    out = model(input_ids=T(g["input_ids"]).to(DEV), attention_mask=T(g["attention_mask"]).to(DEV), labels=T(g["labels"]).to(DEV),
                images=T(g["images"]).to(DEV).bfloat16())
    l_hip, l16, l32 = float(out.loss.detach()), float(g["loss"]), float(g32["loss"])
    rows = T(g["hidden_rows"])
    hs = out.hidden_states[0].float().cpu()[rows]
    e_hip, e_ref = rel(hs, T(g32["hidden"])), rel(T(g["hidden"]), T(g32["hidden"]))
    far = rows >= 2048
    e_far, e_far_ref = rel(hs[far], T(g32["hidden"])[far]), rel(T(g["hidden"])[far], T(g32["hidden"])[far])
    print(f"\n   [rope31 long] loss hip={l_hip:.5f} reference bf16={l16:.5f} fp32={l32:.5f}; hidden rel err vs fp32 hip={e_hip:.3e} "
          f"reference-bf16={e_ref:.3e} (rows >= 2048: {e_far:.3e} / {e_far_ref:.3e}); default-RoPE distance {float(g32['default_rope_hidden_rel']):.2f}")
    assert abs(l_hip - l32) <= max(1.5 * abs(l16 - l32), 1e-3 * abs(l32)), (l_hip, l16, l32)
    assert e_hip <= 1.5 * e_ref and e_far <= 1.5 * e_far_ref
    out.loss.backward()
    n = 0
    for k in g32.files:
        if not k.startswith("grad::"):
            continue
        p = dict(model.named_parameters())[k[6:]]
        got = grad_summary(p.grad)
        e_h, e_r = rel(got[1:], T(g32[k])[1:]), rel(T(g[k])[1:], T(g32[k])[1:])
        assert e_h <= max(2.0 * e_r, 3.3e-2), (k, e_h, e_r)
        n += 1
    assert n >= 20


def test_config_fields_of_real_checkpoints_are_supported_or_refused_by_name():
    """LLaMA-3.1 rope_scaling constructs (it raised NotImplementedError through round 5); dynamic / yarn RoPE, attention_bias and mlp_bias
    are refused at construction, by name."""
    from metamorph_amd.factory import build_model
    base = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, vocab_size=1000,
                rms_norm_eps=1e-5, rope_theta=500000.0, max_position_embeddings=131072)
    geo = dict(num_hidden_layers=1, intermediate_size=144, image_size=56)
    r31 = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
    m = build_model(dict(base, rope_scaling=r31), geo, num_image_tokens=4, max_length=64, device=DEV)
    assert m.model.rope.rope_type == "llama3"
    cos, _ = m.model.rope_tables(4096, DEV)
    cos_d, _ = build_model(base, geo, num_image_tokens=4, max_length=64, device=DEV).model.rope_tables(4096, DEV)
    assert not torch.equal(cos, cos_d) and torch.equal(cos[:, :2], cos_d[:, :2])     # the two shortest wavelengths are kept, long ones stretched
    for bad in ({"rope_type": "dynamic", "factor": 2.0}, {"rope_type": "yarn", "factor": 2.0}):
        with pytest.raises(NotImplementedError, match=bad["rope_type"]):
            build_model(dict(base, rope_scaling=bad), geo, num_image_tokens=4, max_length=64)
    for field in ("attention_bias", "mlp_bias"):
        with pytest.raises(NotImplementedError, match="bias"):
            from metamorph_amd.model import MetaMorphConfig, MetaMorphLlamaForCausalLM
            MetaMorphLlamaForCausalLM(MetaMorphConfig(**base, **{field: True}))


def test_wide_decode_fused_reduce_launches_give_the_same_bits(monkeypatch):
    """More than 16 sequences per step: RoPE + cache append, both RMSNorms and SwiGLU folded into the reduce launches of the split projections
    (nine launches per layer) against the same step with them as launches of their own (thirteen): every score of every step bit for bit,
    on widths that ARE split (hidden 1024 / intermediate 2048: two K slices) -- and fewer launches counted."""
    import metamorph_amd.functional as F
    cfg = tiny_cfg(hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=2, num_image_tokens=4)   # d = 128
    model = hip_model(cfg, init_state_dict(cfg, seed=5, dtype=torch.bfloat16)).eval()
    B = 20
    g = torch.Generator().manual_seed(3)
    lens = [9 + 2 * b for b in range(B)]
    n = max(lens)
    ids = torch.randint(3, 127000, (B, n), generator=g)
    mask = torch.zeros(B, n, dtype=torch.bool)
    for b, L in enumerate(lens):
        mask[b, n - L:] = True
    ids[~mask] = 128001
    kw = dict(use_customize_greedy=False, do_sample=False, max_new_tokens=6, eos_token_id=128009, pad_token_id=128001, return_dict_in_generate=True,
              output_scores=True)
    monkeypatch.setitem(F.VARIANTS, "decode_graph", False)
    monkeypatch.setitem(F.VARIANTS, "prefill_fused", False)      # (the prompt passes would add their own fused launches to the counts)
    runs, counts = {}, {}
    names = ("gemm_splitk", "gemm_splitk_norm", "gemm_splitk_swiglu", "gemm_splitk_rope_append", "rmsnorm_fwd", "swiglu_fwd", "rope_kv_append_", "attn_decode")
    for fused in (True, False):
        seen = []
        with monkeypatch.context() as mp:
            mp.setitem(F.VARIANTS, "decode_wide_fused", fused)
            for name in names:
                orig = getattr(F.ops, name)
                mp.setattr(F.ops, name, lambda *a, _o=orig, _n=name, **k: (seen.append(_n), _o(*a, **k))[1])
            runs[fused] = model.generate(inputs=ids.to(DEV), attention_mask=mask.to(DEV), **kw)
        counts[fused] = {nm: seen.count(nm) for nm in names}
    assert torch.equal(runs[True].sequences, runs[False].sequences)
    for step, (x, y) in enumerate(zip(runs[True].scores, runs[False].scores)):
        assert torch.equal(x, y), (step, float((x - y).abs().max()))
    steps, NL = len(runs[True].scores) - 1, cfg.num_hidden_layers
    assert counts[True]["gemm_splitk_rope_append"] == steps * NL and counts[True]["gemm_splitk_swiglu"] == steps * NL
    assert counts[True]["gemm_splitk_norm"] == steps * (2 * NL - 1) and counts[True]["rope_kv_append_"] == 0
    assert counts[False]["gemm_splitk_norm"] == 0 and counts[False]["rope_kv_append_"] == steps * NL
    print(f"\n   B={B}, {steps} cached steps: fused {counts[True]}; unfused {counts[False]}")


@pytest.mark.parametrize("B", [3, 8, 11, 19, 35])
def test_batched_decode_is_one_pass_and_equals_every_sequence_alone(B, monkeypatch):
    """The cached step of a batch (reference: the whole batch goes to ONE forward per step, metamorph_llama.py:711-717) takes all B rows
    through every decoder layer in one pass -- 5 launches per layer (7 from five rows on: the norms run on their own) for B <= 16 (the GEMV
    kernels' M); beyond 16 rows (round 6) still ONE pass: the projections take the split-K GEMM, attention one launch per layer -- not
    B passes, and not ceil(B / 16) passes; and every sequence gets what it gets alone: prompts of B different lengths (left-padded batch), eight greedy steps, per-step
    logits of every row against the same prompt decoded on its own (accumulation-order accuracy; argmax ids equal wherever the top two
    logits are further apart than that accuracy)."""
    import metamorph_amd.functional as F
    cfg = tiny_cfg(hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=1, num_image_tokens=4)   # d = 128
    model = hip_model(cfg, init_state_dict(cfg, seed=21, dtype=torch.bfloat16)).eval()
    g = torch.Generator().manual_seed(B)
    lens = [12 + 3 * b for b in range(B)]
    n = max(lens)
    ids = torch.randint(3, 127000, (B, n), generator=g)
    mask = torch.zeros(B, n, dtype=torch.bool)
    for b, L in enumerate(lens):
        mask[b, n - L:] = True
    ids[~mask] = 128001
    launches = []
    for name in ("gemv_rope_append", "attn_decode", "gemv", "gemv_swiglu"):
        orig = getattr(F.ops, name)
        monkeypatch.setattr(F.ops, name, lambda *a, _o=orig, _n=name, **k: (launches.append(_n), _o(*a, **k))[1])
    monkeypatch.setitem(F.VARIANTS, "decode_graph", False)       # eager launches: countable (the graph replays the same sequence)
    kw = dict(use_customize_greedy=False, do_sample=False, max_new_tokens=8, eos_token_id=128009, pad_token_id=128001, return_dict_in_generate=True,
              output_scores=True)
    out = model.generate(inputs=ids.to(DEV), attention_mask=mask.to(DEV), **kw)
    per_step = 5 * cfg.num_hidden_layers * ((B + 15) // 16) + 1                                   # counted launches: the GEMVs + attention (+ lm_head)
    steps = len(out.scores) - 1
    n_dec = len(launches)
    print(f"\n   B={B}: {n_dec} decode-shape launches over {steps} cached steps (+ prompt pass)")
    assert n_dec <= (steps + 1) * per_step + 2 * B, (n_dec, steps, per_step)      # one pass per step for the batch, not one per row
    if B > 16:                                                   # one attention launch per layer and step: the batch was not chunked
        assert launches.count("attn_decode") == steps * cfg.num_hidden_layers, (launches.count("attn_decode"), steps)
    monkeypatch.undo()
    for b in range(B):
        alone = model.generate(inputs=ids[b:b + 1, n - lens[b]:].to(DEV), **kw)
        for step, (x, y) in enumerate(zip(alone.scores, out.scores)):
            # (alone: one row = an fp32 fma chain; in the batch: 3 .. 4 rows = v_dot2c_f32_bf16, 5 .. 16 = MFMA -- the same products in another fp32
            # summation order, re-rounded to bf16 after every projection: differences of a bf16 step of the hidden state, i.e. ~1e-2 on a logit)
            assert torch.allclose(x[0], y[b], rtol=2e-2, atol=2e-2), (b, step, float((x[0] - y[b]).abs().max()))
            if int(x[0].argmax()) != int(y[b].argmax()):
                # a random-init model has near-ties: the two runs may pick different ids only where the top two logits are closer than
                # that accuracy, and from there on they decode different sequences (nothing further to compare)
                top = torch.topk(y[b].float(), 2).values
                assert float(top[0] - top[1]) < 2e-2, (b, step, float(top[0] - top[1]))
                break
        else:
            assert alone.sequences[0].tolist() == out.sequences[b].tolist(), (b, alone.sequences[0].tolist(), out.sequences[b].tolist())


def test_decode_step_graph_switches_its_attention_bound_when_a_sequence_passes_1024_rows():
    """mm355_attn_decode's bound on the cached lengths is a launch parameter (1024 while every sequence fits one key group: the
    one-workgroup-per-head form; the capacity afterwards).  DecodeStepGraph keeps one captured step per bound and picks by the lengths the
    host knows: a batch whose longest sequence grows from 1018 to 1030 rows replays the short graph, then the long one, and every step's
    rows equal the eager step under the capacity bound bit for bit."""
    import metamorph_amd.functional as F
    cfg = tiny_cfg(hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=1, num_image_tokens=4)   # d = 128
    model = hip_model(cfg, init_state_dict(cfg, seed=5, dtype=torch.bfloat16)).eval()
    cap, B, h = 1100, 3, 512
    _, meta = model._decode_meta(16)
    cos, sin = model.model.rope_tables(cap, DEV)
    meta.cos, meta.sin = cos, sin
    g = torch.Generator().manual_seed(3)
    k0 = (torch.randn(cfg.num_hidden_layers, B, cap, meta.Hkv * meta.d, generator=g) * 0.5).bfloat16().to(DEV)
    v0 = (torch.randn(cfg.num_hidden_layers, B, cap, meta.Hkv * meta.d, generator=g) * 0.5).bfloat16().to(DEV)
    rows = [(torch.randn(B, h, generator=g) * 0.5).bfloat16().to(DEV) for _ in range(12)]
    outs = []
    with torch.no_grad():
        for graph in (True, False):
            kv = F.KVCache(cfg.num_hidden_layers, cap, meta.Hkv * meta.d, DEV, Hq=meta.Hq, d=meta.d, batch=B)
            kv.k.copy_(k0); kv.v.copy_(v0)
            kv.set_lengths([1018, 1000, 7])
            if graph:
                st = F.DecodeStepGraph(model.model.layers, meta, kv, cos, sin, h, DEV)
                assert st.capture_ok
                outs.append([st.step(r).clone() for r in rows])
                assert sorted(st.graphs) == [F.SHORT_KV, cap], sorted(st.graphs)
            else:
                outs.append([F.decoder_decode_row(r, model.model.layers, meta, kv, cos, sin, kv_bound=cap).clone() for r in rows])
            assert kv.lengths == [1030, 1012, 19]
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


def test_forward_with_host_mirrors_does_not_touch_the_device_for_its_plan():
    """The splice plan is host integer work; with the batch's host originals registered as mirrors (metamorph_amd.hostmirror: what
    MetaMorphTrainer._prepare_inputs and bench.py do when they move a batch to the device) `forward` builds it without copying ids / labels /
    mask back -- no device -> host synchronisation per step (round 4: one `.cpu()` per tensor and step) -- and returns the same bits."""
    from metamorph_amd import hostmirror as HM
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, init_state_dict(cfg, seed=11, dtype=torch.bfloat16))
    model.train()
    ids, msk, lab, img = T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), T(g["images"]).to(DEV).bfloat16()
    def poison(value):                                            # the caching allocator hands these blocks to the next torch.empty calls
        junk = [torch.full((n,), value, device=DEV, dtype=torch.bfloat16) for n in (1 << 24, 1 << 20, 1 << 16, 1 << 12, 1 << 12, 1 << 10)]
        del junk
    poison(float("nan"))
    plain = model(input_ids=ids.to(DEV), attention_mask=msk.to(DEV), labels=lab.to(DEV), images=img)
    assert bool(torch.isfinite(plain.hidden_states.float()).all()), "a forward buffer is read before it is written"
    poison(3.0)
    s0 = dict(HM.STATS)
    mirrored = model(input_ids=HM.to_device(ids, DEV), attention_mask=HM.to_device(msk, DEV), labels=HM.to_device(lab, DEV), images=img)
    assert HM.STATS["sync"] == s0["sync"] and HM.STATS["mirror"] >= s0["mirror"] + 3, (s0, HM.STATS)
    cpu_in = model(input_ids=ids, attention_mask=msk, labels=lab, images=img)                 # CPU tensors are their own mirror
    assert HM.STATS["sync"] == s0["sync"]
    for name, out in (("mirrored", mirrored), ("cpu inputs", cpu_in)):
        # (the loss SCALAR is a sum of per-row terms by fp32 atomicAdd across workgroups: its last bit depends on the order of arrival -- seen
        # once in a full-suite run: 11.9583979 vs 11.9583998 --; hidden states and gradients have no atomics on their path)
        assert abs(float(out.loss) - float(plain.loss)) <= 4e-6 * abs(float(plain.loss)), (name, float(out.loss), float(plain.loss))
        dh = (out.hidden_states.float() - plain.hidden_states.float()).abs()
        assert torch.equal(out.hidden_states, plain.hidden_states), (name, float(dh.max()), torch.nonzero(dh.amax(-1))[:8].tolist(), msk.tolist())
    mirrored.loss.backward()
    assert HM.STATS["sync"] == s0["sync"]


def test_bench_emits_the_driver_contract_on_device():
    """`python bench.py --layers 2 --vit-layers 2 --steps 2 --warmup 1 --batch 2` really runs (a 2-layer debug geometry, NOT the headline
    config) and its ONE stdout line carries the driver's contract: the required keys, value = tokens of the timed steps / their wall time,
    `roofline` over ALL GEMM launches with the plain / fused split and the HBM-bound block, `cpu_baseline` with cores and sample."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--layers", "2", "--vit-layers", "2", "--steps", "2", "--warmup", "1", "--batch", "2"],
                       capture_output=True, text=True, timeout=900, cwd=repo)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["unit"] == "tokens/s" and r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "bf16"
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and "workload" in r["config"] and "model" not in r["config"] and "synthetic" in r["data"]
    tokens = r["config"]["global_batch"] * r["config"]["seq_len"]
    assert abs(r["value"] - tokens / (r["ms_per_step"] * 1e-3)) <= 2e-3 * r["value"]
    roof = r["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["launches"] == roof["plain"]["launches"] + roof["fused_mlp"]["launches"]
    fl = roof["plain"]["algorithmic_flops_per_step"] + roof["fused_mlp"]["algorithmic_flops_per_step"]
    assert abs(roof["algorithmic_flops_per_step"] - fl) <= 1e-6 * fl
    assert {"rmsnorm_fwd", "transpose", "adamw_shard"} <= set(roof["hbm_bound"]) and all(0 < v["achieved_tb_s"] < 8.5 for v in roof["hbm_bound"].values())
    cb = r["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert r["cpu_baseline_c1"]["fp32"]["steps_timed"] >= 3 and r["cpu_baseline_c1"]["fp32"]["warmup_steps"] >= 1
    assert r["batch_pool"] >= 8 and math.isfinite(r["loss"])


# ------------------------------------------------------------------ BASELINE configs[0] geometry class (TinyLlama: d = 64, GQA 8:1)
D64_HIDDEN_TOL, D64_GRAD_TOL = 9.5e-3, 2.2e-2          # 1.5 x measured (6.3e-3, 1.44e-2)


def test_e2e_head_dim_64_gqa8_against_oracle():
    """TinyLlama-style attention geometry (head size 64, eight query heads per KV head) goes through the generic attention
    kernels (attn2) instead of the d = 128 LDS-DMA ones; loss and gradients against the CPU oracle on the same weights (fp32)."""
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))               # inputs only (ids / labels / images)
    cfg = tiny_cfg(hidden_size=512, intermediate_size=768, num_attention_heads=8, num_key_value_heads=1, num_image_tokens=4)
    seed = 11
    model = hip_model(cfg, init_state_dict(cfg, seed=seed, dtype=torch.bfloat16))
    model.train()
    batch = dict(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]))
    out = model(input_ids=batch["input_ids"].to(DEV), attention_mask=batch["attention_mask"].to(DEV), labels=batch["labels"].to(DEV),
                images=batch["images"].to(DEV).bfloat16())
    sd = {k: v.bfloat16().float() for k, v in init_state_dict(cfg, seed=seed).items()}     # the same bf16-rounded weights, fp32 math
    for k, v in sd.items():
        v.requires_grad_("vision_tower" not in k and "vision_proj" not in k)
    ref = oracle_forward(sd, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"], return_logits=False)
    got, want = float(out.loss.detach()), float(ref["loss"].detach())
    print(f"\n   d=64 GQA8 loss hip={got:.5f} oracle={want:.5f}")
    assert abs(got - want) <= 1e-3 * abs(want)
    mask = ref["attention_mask"]
    e_hid = rel(out.hidden_states.float().cpu()[mask], ref["hidden_states"].detach()[mask])
    print(f"   hidden rel err {e_hid:.3e}")
    assert e_hid <= D64_HIDDEN_TOL
    out.loss.backward()
    ref["loss"].backward()
    params = dict(model.named_parameters())
    n, worst = 0, (0.0, "")
    for k, v in sd.items():
        if v.grad is None or k not in params:
            continue
        e = rel(params[k].grad, v.grad)
        worst = max(worst, (e, k))
        assert e <= D64_GRAD_TOL, (k, e)
        n += 1
    print(f"   {n} gradient tensors, worst rel err {worst[0]:.3e} ({worst[1]})")
    assert n >= 20


# ------------------------------------------------------------------ row N4: trainable vision tower (freeze_vision=False)
TOWER_GRAD_TOL = 1.8e-2                                    # 1.5 x measured (1.15e-2, layers.1.q_proj)


def test_trainable_vision_tower_gradients_against_oracle():
    """reference siglip_encoder.py:138-139 (`torch.set_grad_enabled(not self.freeze_vision)`): with the tower unfrozen the loss
    back-propagates through mm_projector, the 729 -> T reduction + L2 norm and every SigLIP encoder layer; gradients of all tower
    parameters (patch embedding, position embedding, LayerNorms, q/k/v/out, fc1/fc2) against autograd through the CPU oracle."""
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    seed = 21
    model = hip_model(cfg, init_state_dict(cfg, seed=seed, dtype=torch.bfloat16))
    model.train()
    tower = model.get_model().vision_tower
    tower.freeze_vision = False
    for n, p in tower.named_parameters():
        p.requires_grad_("post_layernorm" not in n)
    batch = dict(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]))
    out = model(input_ids=batch["input_ids"].to(DEV), attention_mask=batch["attention_mask"].to(DEV), labels=batch["labels"].to(DEV),
                images=batch["images"].to(DEV).bfloat16())
    sd = {k: v.bfloat16().float() for k, v in init_state_dict(cfg, seed=seed).items()}
    for k, v in sd.items():
        v.requires_grad_("vision_proj" not in k and "post_layernorm" not in k)
    ref = oracle_forward(sd, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"], return_logits=False,
                         train_vision=True)
    got, want = float(out.loss.detach()), float(ref["loss"].detach())
    assert abs(got - want) <= 1e-3 * abs(want), (got, want)
    out.loss.backward()
    ref["loss"].backward()
    params = dict(model.named_parameters())
    n_tower, worst = 0, (0.0, "")
    for k, v in sd.items():
        if "vision_tower" not in k or v.grad is None:
            continue
        assert params[k].grad is not None, k
        if k.endswith("k_proj.bias"):
            # softmax is invariant to a shift of all keys along q: the exact gradient is ZERO (the oracle shows fp32 noise);
            # ours must be bf16 noise next to the q bias gradient
            qb = float(sd[k.replace("k_proj", "q_proj")].grad.norm())
            assert float(v.grad.norm()) <= 1e-4 * max(qb, 1e-12) and float(params[k].grad.float().norm()) <= 5e-2 * qb, k
        else:
            e = rel(params[k].grad, v.grad)
            worst = max(worst, (e, k))
            assert e <= TOWER_GRAD_TOL, (k, e)
        n_tower += 1
    print(f"\n   trainable tower: loss hip={got:.5f} oracle={want:.5f}; {n_tower} tower gradient tensors checked, worst rel err {worst[0]:.3e} ({worst[1]})")
    assert n_tower == 3 + 16 * cfg.v_layers
    # the frozen default still refuses nothing and produces no tower gradients
    model2 = hip_model(cfg, init_state_dict(cfg, seed=seed, dtype=torch.bfloat16))
    assert all(not p.requires_grad for p in model2.get_model().vision_tower.parameters())


# ------------------------------------------------------------------ A3 directly: HIP tower vs the reference-recorded tower outputs
@pytest.mark.parametrize("Timg", [4, 16])
def test_tower_output_matches_reference_recorded(Timg):
    """tests/golden/a3_tower_T*_{f32,bf16}.npz hold the reference's own SiglipVisionTower outputs (siglip_encoder.py:138-213):
    hidden_states[-1] of the HF encoder (strided sample), the reduced + L2-normalised features, the projector output and the
    detached target.  The HIP tower (im2col + GEMM patch embedding, LayerNorm, attn2 d = 72, tanh-GELU epilogues,
    bilinear_l2norm) must be as close to the reference's fp32 run as the reference's bf16 run is (x2)."""
    g16 = np.load(os.path.join(GOLDEN, f"a3_tower_T{Timg}_bf16.npz"))
    g32 = np.load(os.path.join(GOLDEN, f"a3_tower_T{Timg}_f32.npz"))
    cfg = tiny_cfg(num_image_tokens=Timg)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g16["seed"]), dtype=torch.bfloat16)).eval()
    images = T(g16["images"]).to(DEV).bfloat16()
    tower = model.get_model().vision_tower
    with torch.no_grad():
        raw = tower.vision_tower.forward_features(images, tower.select_layer)
        feat = tower(images)
        proj, tgt = model.encode_images(images)
    for name, got, key in (("raw_hidden", raw[:, :, ::8], "raw_hidden"), ("features", feat, "features"),
                           ("projected", proj[:, :, ::4], "projected"), ("target", tgt[:, :, ::16], "target")):
        e_hip, e_ref = rel(got, T(g32[key])), rel(T(g16[key]), T(g32[key]))
        print(f"\n   tower T={Timg} {name}: rel err vs reference fp32 hip={e_hip:.3e} reference-bf16={e_ref:.3e}")
        assert got.shape == tuple(g32[key].shape)
        assert e_hip <= max(1.5 * e_ref, 4e-3), (name, e_hip, e_ref)   # measured: at or below the reference's own bf16 distance
    assert torch.equal(tgt, feat)                                  # the regression target is the detached tower output (A4)
    n = feat.float().norm(dim=-1)
    assert float((n - 1).abs().max()) < 1e-2                       # normalize_vision


def test_tower_select_layer_and_zero_token_branches_match_reference_recorded():
    """tests/golden/a3sel_tower_*.npz: `mm_vision_select_layer = -2` on a 3-layer tower (hidden_states[-2], siglip_encoder.py:129-131) and
    `num_image_tokens = -1` (zeros of the un-reduced shape, :146-148), as the reference's tower returned them."""
    g16, g32 = (np.load(os.path.join(GOLDEN, f"a3sel_tower_{t}.npz")) for t in ("bf16", "f32"))
    cfg = tiny_cfg(num_image_tokens=4, v_layers=3)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g16["seed"]), dtype=torch.bfloat16)).eval()
    images = T(g16["images"]).to(DEV).bfloat16()
    tower = model.get_model().vision_tower
    with torch.no_grad():
        tower.select_layer = -2
        raw = tower.vision_tower.forward_features(images, tower.select_layer)
        feat = tower(images)
        last = tower.vision_tower.forward_features(images, -1)
        tower.select_layer, tower.image_token_len = -1, -1
        zeros = tower(images)
    assert not torch.equal(raw, last)
    for name, got, key in (("hidden_states[-2]", raw[:, :, ::8], "raw_hidden_m2"), ("features", feat, "features_m2")):
        e_hip, e_ref = rel(got, T(g32[key])), rel(T(g16[key]), T(g32[key]))
        print(f"\n   tower select_layer=-2 {name}: rel err vs reference fp32 hip={e_hip:.3e} reference-bf16={e_ref:.3e}")
        assert got.shape == tuple(g32[key].shape)
        assert e_hip <= max(1.5 * e_ref, 4e-3), (name, e_hip, e_ref)
    assert list(zeros.shape) == g32["tokens_minus1_shape"].tolist() and zeros.dtype == torch.bfloat16 and float(zeros.abs().max()) == 0.0
    with pytest.raises(IndexError):
        tower.vision_tower.forward_features(images, -5)          # 3 layers: hidden_states has 4 entries


@pytest.mark.parametrize("stem", ["r3_textonly", "r3_textonly_left"])
def test_text_only_forward_without_images_matches_reference_recorded(stem):
    """forward(images=None) (r3_textonly_*.npz, recorded from the reference): the splice returns early (metamorph_arch.py:184-191), loss = CE
    alone, no gradient reaches the projector / vision head."""
    g, g32 = (np.load(os.path.join(GOLDEN, f"{stem}_{t}.npz")) for t in ("bf16", "f32"))
    # "_left": the same batch left-padded under tokenizer_padding_side = "left" (HF: position_ids = arange(L), padding included)
    cfg = tiny_cfg(num_image_tokens=4, tokenizer_padding_side="left" if stem.endswith("left") else "right")
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    model.train()
    ids, lab, msk = T(g["input_ids"]).to(DEV), T(g["labels"]).to(DEV), T(g["attention_mask"]).to(DEV)
    out = model(input_ids=ids, attention_mask=msk, labels=lab, images=None)
    got, ref16, truth = float(out.loss.detach()), float(g["loss"]), float(g32["loss"])
    valid = T(g["attention_mask"]).bool()
    e_h, e_r = rel(out.hidden_states.float().cpu()[valid], T(g32["hidden"])[valid]), rel(T(g["hidden"])[valid], T(g32["hidden"])[valid])
    print(f"\n   text only, images=None: loss hip={got:.6f} reference bf16={ref16:.6f} fp32={truth:.6f}; hidden rel err hip={e_h:.3e} reference-bf16={e_r:.3e}")
    assert abs(got - truth) <= 1e-3 * abs(truth) and abs(got - ref16) <= 1e-3 * abs(ref16)
    assert e_h <= max(1.5 * e_r, 1.2e-2)
    out.loss.backward()
    params = dict(model.named_parameters())
    for n in g["params_without_grad"].tolist():
        assert params[n].grad is None or float(params[n].grad.float().abs().max()) == 0.0, n
    worst = 0.0
    for k in g32.files:
        if k.startswith("grad::"):
            got_g = grad_summary(params[k[6:]].grad)
            nz = int((T(g32[k])[1:] != 0).sum())
            if nz > 4:
                e = rel(got_g[1:], T(g32[k])[1:])
                worst = max(worst, e)
                assert e <= max(3.0 * rel(T(g[k])[1:], T(g32[k])[1:]), 3.3e-2), (k, e)
            assert abs(float(got_g[0]) - float(g32[k][0])) <= 3e-2 * float(g32[k][0]), k
    print(f"   gradients: worst rel err vs reference fp32 {worst:.3e}")


# ------------------------------------------------------------------ BASELINE configs[0] at its real widths
def _fullwidth_check(cfg, ids, labels, mask, images, seed, *, grad_tol, hidden_tol, what, check_embed_grad=True):
    """HIP model (bf16) vs the CPU oracle in fp32 on the SAME bf16-rounded weights: integer outputs bit-exact, loss, valid hidden
    rows, and the gradient of every trainable tensor."""
    sd16 = init_state_dict(cfg, seed=seed, dtype=torch.bfloat16, fast_big=True)
    model = hip_model(cfg, sd16)
    model.train()
    out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), images=images.to(DEV).bfloat16())
    plan = model.prepare_inputs_labels_for_multimodal(ids.to(DEV), None, mask.to(DEV), None, labels.to(DEV), images.to(DEV).bfloat16())
    sd = {k: v.float() for k, v in sd16.items()}
    for k, v in sd.items():
        # embed_tokens: autograd through the oracle's row-by-row splice costs one [V, h] zero tensor PER ROW (minutes at V = 128258);
        # its gradient is checked on the device instead (below), all other trainable tensors against autograd
        v.requires_grad_("vision_tower" not in k and "vision_proj" not in k and (check_embed_grad or k != "model.embed_tokens.weight"))
    ref = oracle_forward(sd, cfg, ids, mask, labels, images.bfloat16().float(), return_logits=False, ce_rows_only=True)
    # integer bookkeeping of the splice: bit-exact
    assert torch.equal(plan[5].cpu(), ref["labels"]) and torch.equal(plan[6].cpu(), ref["image_positions"])
    assert torch.equal(plan[2].cpu().bool(), ref["attention_mask"])
    got, want = float(out.loss.detach()), float(ref["loss"].detach())
    print(f"\n   {what}: loss hip={got:.5f} oracle-fp32={want:.5f} lang={model.loss_language:.5f}/{ref['loss_language']:.5f} "
          f"img={model.loss_image_ar:.5f}/{ref['loss_image_ar']:.5f}")
    assert abs(got - want) <= 1e-3 * abs(want), (got, want)
    assert abs(model.loss_language - ref["loss_language"]) <= 1e-3 * abs(want)
    assert abs(model.loss_image_ar - ref["loss_image_ar"]) <= 1e-3
    valid = ref["attention_mask"]
    e = rel(out.hidden_states.float().cpu()[valid], ref["hidden_states"].detach()[valid])
    print(f"   hidden rel err {e:.3e}")
    assert e <= hidden_tol
    out.loss.backward()
    ref["loss"].backward()
    params = dict(model.named_parameters())
    n, worst = 0, (0.0, "")
    for k, v in sd.items():
        if v.grad is None or k not in params:
            continue
        assert params[k].grad is not None, k
        e = rel(params[k].grad, v.grad)
        worst = max(worst, (e, k))
        assert e <= grad_tol, (k, e)
        n += 1
    print(f"   {n} gradient tensors, worst rel err {worst[0]:.3e} ({worst[1]})")
    if not check_embed_grad:
        # d loss / d embed_tokens = scatter-add of d inputs_embeds over the token ids: rows of ids that never occur stay exactly zero,
        # and the total over rows equals the sum of the text rows of d inputs_embeds -- checked against the oracle's d inputs_embeds
        ge = params["model.embed_tokens.weight"].grad
        used = torch.unique(ids[ids >= 0])
        unused = torch.ones(ge.shape[0], dtype=torch.bool)
        unused[used] = False
        assert float(ge[unused.to(ge.device)].float().abs().max()) == 0
    return n


def test_configs0_tinyllama_real_widths_against_oracle():
    """BASELINE configs[0] at its real widths -- TinyLlama-1.1B geometry (h 2048, 32 query / 4 KV heads of size 64, I 5632,
    V 32002 with <image_start> = 32000: the `image_start_id` the reference hard-codes as 128256 is a config field here), two decoder
    layers, SO400M/14-384 tower geometry (729 patches -> 256 tokens) with two layers; 1 prompt image + 128 text ids
    plus a second sample whose image is answer-side, so that the label rule with the re-based start id is exercised."""
    cfg = OracleConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=4,
                       vocab_size=32002, rope_theta=10000.0, v_layers=2, num_image_tokens=256, tokenizer_model_max_length=2048,
                       image_start_id=32000)
    g = torch.Generator().manual_seed(1234)
    n_ids = 129
    ids = torch.randint(3, 31999, (2, n_ids), generator=g)
    ids[:, 0] = 1
    ids[:, 21], ids[:, 22], ids[:, 23] = 32000, -200, 32001
    labels = torch.full_like(ids, -100)
    labels[0, -64:] = ids[0, -64:]                              # image-QA sample: the last 64 positions are supervised
    labels[1, 18:] = ids[1, 18:]                                # generation sample: supervised from before <image_start>
    labels[1, 22] = -200
    mask = torch.ones_like(ids, dtype=torch.bool)
    images = torch.randn(2, 3, 384, 384, generator=g)
    # measured (round 3): hidden 7.5e-3, worst gradient 1.7e-2 (layers.1.q_proj) -> 1.5 x
    n = _fullwidth_check(cfg, ids, labels, mask, images, seed=31, grad_tol=2.6e-2, hidden_tol=1.2e-2, what="configs[0] TinyLlama widths")
    assert n >= 20


# (BASELINE configs[2] -- L = 4096, eight interleaved frames at LLaMA-3-8B widths -- is test_configs2_shape_8_frames_seq4096_eight_layers_*
#  below: eight decoder layers on the same shape.  Its two-layer twin on the plain oracle, 52 s of host GEMMs for a strict subset of that
#  coverage, was retired in round 6 to keep the -m gpu suite inside the driver's step limit.)


# ------------------------------------------------------------------ padding-free rows (round 6)
@pytest.mark.parametrize("side", ["right", "left"])
def test_padding_free_rows_equal_the_padded_path(side):
    """Ragged batch (lengths 700 / 333 / 90 of L = 700: 47 % padding): the decoder on COMPACT rows (mm355_compact_rows, the default from 8 %
    padding) against the same model on the padded layout (mm355_compact_rows = False).  Every row-wise kernel computes a row from that row
    alone, and attention sees the same padded q|k|v either way: loss, loss_language / loss_image_ar, logits-free hidden rows and the input
    gradient are compared BIT FOR BIT on the valid rows; weight gradients contract over rows (zero rows removed = another fp32 summation
    order inside the MFMAs) and are held to 2e-3 of their norm.  Both padding sides; 1 536 instead of 2 100 decoder rows."""
    cfg = OracleConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                       vocab_size=32002, v_layers=1, v_intermediate=144, v_image=56, num_image_tokens=16, tokenizer_model_max_length=1024,
                       image_start_id=32000, tokenizer_padding_side=side)
    g = torch.Generator().manual_seed(21)
    lens = [685, 318, 75]                                         # ids per sample; + 15 image rows each = 700 / 333 / 90 spliced rows
    n_ids = max(lens)
    ids = torch.zeros((3, n_ids), dtype=torch.long)
    labels = torch.full((3, n_ids), -100, dtype=torch.long)
    mask = torch.zeros((3, n_ids), dtype=torch.bool)
    for b, n in enumerate(lens):
        row = torch.randint(3, 31999, (n,), generator=g)
        row[0] = 1
        p = 20 if b != 2 else n - 4                               # samples 0, 1: prompt-side image; sample 2: answer-side image at its end
        row[p], row[p + 1], row[p + 2] = 32000, -200, 32001
        lab = torch.full((n,), -100, dtype=torch.long)
        lab[n // 2:] = row[n // 2:]
        sl = slice(n_ids - n, n_ids) if side == "left" else slice(0, n)
        ids[b, sl], labels[b, sl], mask[b, sl] = row, lab, True
    images = torch.randn(3, 3, 56, 56, generator=g)
    sd = init_state_dict(cfg, seed=19, dtype=torch.bfloat16)
    res = {}
    for compact in (False, "auto"):
        model = hip_model(cfg, sd)
        model.train()
        model.config.mm355_compact_rows = compact
        out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), images=images.to(DEV).bfloat16())
        out.loss.backward()
        torch.cuda.synchronize()
        res[compact] = dict(loss=out.loss.detach().clone(), lang=model.loss_language, img=model.loss_image_ar, hid=out.hidden_states.detach().clone(),
                            rows=model._decoder_rows, grads={n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
        del model
    a, b = res[False], res["auto"]
    assert a["rows"] == (2100, 2100) and b["rows"] == (1280, 2100), (a["rows"], b["rows"])       # 700 + 333 + 90 = 1123 -> 1280 (whole 256-row tiles)
    assert torch.equal(a["loss"], b["loss"]) and a["lang"] == b["lang"] and a["img"] == b["img"], (float(a["loss"]), float(b["loss"]))
    assert torch.equal(a["hid"], b["hid"])                        # every row, padding included (zeros on both paths)
    worst = (0.0, "")
    for n, ga in a["grads"].items():
        e = float((b["grads"][n] - ga).norm() / ga.norm().clamp_min(1e-30))
        worst = max(worst, (e, n))
        assert e <= 2e-3, (n, e)
    same = sum(int(torch.equal(a["grads"][n], b["grads"][n])) for n in a["grads"])
    print(f"\n   padding-free vs padded ({side} padding): loss / hidden rows bit-identical; {same} of {len(a['grads'])} gradient tensors bit-identical, "
          f"worst rel diff {worst[0]:.2e} ({worst[1]})")


def test_padding_free_row_count_modes():
    """The compact row count across steps of one batch shape: "auto" / True keep the high-water mark of the batches seen (constant tensor
    sizes), "exact" follows every batch (256-row granule); either way the loss is that of the padded layout, bit for bit."""
    cfg = OracleConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       vocab_size=32002, v_layers=1, v_intermediate=144, v_image=56, num_image_tokens=16, tokenizer_model_max_length=1024,
                       image_start_id=32000)
    sd = init_state_dict(cfg, seed=23, dtype=torch.bfloat16)

    def batch(lens, seed):
        g = torch.Generator().manual_seed(seed)
        n_ids = 600                                               # every batch padded to the same 615 spliced rows
        ids = torch.zeros((3, n_ids), dtype=torch.long)
        labels = torch.full((3, n_ids), -100, dtype=torch.long)
        mask = torch.zeros((3, n_ids), dtype=torch.bool)
        for b, n in enumerate(lens):
            row = torch.randint(3, 31999, (n,), generator=g)
            row[0] = 1
            row[5], row[6], row[7] = 32000, -200, 32001
            ids[b, :n], mask[b, :n] = row, True
            labels[b, n // 2:n] = row[n // 2:]
        return dict(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), images=torch.randn(3, 3, 56, 56, generator=g).to(DEV).bfloat16())

    batches = [batch([600, 300, 80], 1), batch([600, 100, 40], 2), batch([600, 500, 400], 3)]      # 1025 / 785 / 1545 valid rows of 1845
    seen = {}
    for mode in (False, True, "exact"):
        model = hip_model(cfg, sd)
        model.config.mm355_compact_rows = mode
        with torch.no_grad():
            seen[mode] = []
            for bt in batches + batches[:1]:
                out = model(**bt)
                # (prompt-side images only: no image-AR rows, `loss` itself is the reference's NaN -- the language loss and the hidden rows carry the comparison)
                seen[mode].append(((model.loss_language, out.hidden_states.clone()), model._decoder_rows[0]))
        del model
    assert [r for _, r in seen[False]] == [1845] * 4
    assert [r for _, r in seen[True]] == [1280, 1280, 1792, 1792], seen[True]       # steps of max(256, ceil256(1845 / 16)) = 256 rows; never shrinks
    assert [r for _, r in seen["exact"]] == [1280, 1024, 1792, 1280], seen["exact"]
    for mode in (True, "exact"):
        assert all(a[0] == b[0] and torch.equal(a[1], b[1]) for (a, _), (b, _) in zip(seen[False], seen[mode])), mode
    assert all(a[0] == a[0] and a[0] > 0 for a, _ in seen[False])   # finite language losses


def test_padding_free_rows_under_gradient_checkpointing_are_bit_identical():
    """Per-layer recompute (`gradient_checkpointing_enable()`, every reference launch script) re-runs a layer's forward kernels on the same
    compact rows: loss and every gradient equal the non-recomputing compact run bit for bit."""
    cfg = OracleConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       vocab_size=32002, v_layers=1, v_intermediate=144, v_image=56, num_image_tokens=16, tokenizer_model_max_length=512,
                       image_start_id=32000)
    g = torch.Generator().manual_seed(3)
    ids = torch.zeros((2, 300), dtype=torch.long)
    labels = torch.full((2, 300), -100, dtype=torch.long)
    mask = torch.zeros((2, 300), dtype=torch.bool)
    for b, n in enumerate((300, 120)):
        row = torch.randint(3, 31999, (n,), generator=g)
        row[0] = 1
        row[n - 4], row[n - 3], row[n - 2] = 32000, -200, 32001
        ids[b, :n], mask[b, :n] = row, True
        labels[b, n // 2:n] = row[n // 2:]
        labels[b, n - 3] = -200
    images = torch.randn(2, 3, 56, 56, generator=g)
    sd = init_state_dict(cfg, seed=29, dtype=torch.bfloat16)
    res = []
    for ckpt in (False, True):
        model = hip_model(cfg, sd)
        model.train()
        if ckpt:
            model.gradient_checkpointing_enable()
        out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), images=images.to(DEV).bfloat16())
        out.loss.backward()
        assert model._decoder_rows[0] < model._decoder_rows[1]             # compact rows were on (28 % padding)
        res.append((out.loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        del model
    assert torch.equal(res[0][0], res[1][0]) and res[0][1].keys() == res[1][1].keys()
    assert all(torch.equal(res[0][1][n], res[1][1][n]) for n in res[0][1])


# ------------------------------------------------------------------ the fp32 oracle evaluated on the device (round 6)
def test_streamed_oracle_on_device_against_reference_recorded_8_layer_run():
    """oracle/ref_stream.full_depth(device="cuda") -- the evaluation the full-width tests below use for their fp32 truth -- DIRECTLY against
    the reference: tests/golden/e2e_multi_frame_T4_ar1_deep8_f32.npz is the reference's own fp32 forward + backward of an 8-layer decoder +
    4-layer tower (the host evaluation passes the same check in tests/test_oracle_stream.py)."""
    from oracle.ref_stream import full_depth
    g = np.load(os.path.join(GOLDEN, "e2e_multi_frame_T4_ar1_deep8_f32.npz"))
    cfg = tiny_cfg(num_image_tokens=4, **json.loads(str(g["cfg_json"])))
    assert cfg.num_hidden_layers == 8 and cfg.v_layers == 4
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    got = full_depth(lambda k: sd[k].detach().clone(), cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), T(g["images"]),
                     grad_layers=(0, 7), device=DEV)
    assert all(not v.is_cuda for v in got["grads"].values()) and not got["hidden_states"].is_cuda       # results come back on the host
    assert abs(got["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    assert abs(got["loss_language"] - float(g["loss_language"])) <= 2e-5 * abs(float(g["loss_language"]))
    assert abs(got["loss_image_ar"] - float(g["loss_image_ar"])) <= 2e-5
    valid = got["attention_mask"]
    torch.testing.assert_close(got["hidden_states"][valid], T(g["hidden"])[valid], rtol=2e-4, atol=2e-5)
    checked = 0
    for name, grad in got["grads"].items():
        torch.testing.assert_close(grad_summary(grad), T(g["grad::" + name]), rtol=5e-4, atol=2e-6)
        checked += 1
    assert checked == 18 + 2 + 4 + 4


def test_streamed_oracle_on_device_equals_host_evaluation():
    """The same fp32 oracle run on the host and on the device at a width where GEMM blocking differs (h 1024, 4 layers, 16 / 4 heads of 64,
    600 + 300 rows, 27 x 27 -> 16 tower tokens): loss to 2e-6, hidden rows to 1e-5, every gradient to 2e-4 -- accumulation-order noise of
    fp32, three orders of magnitude below anything the bf16 comparisons measure."""
    from oracle.ref_stream import full_depth
    cfg = OracleConfig(hidden_size=1024, intermediate_size=2048, num_hidden_layers=4, num_attention_heads=16, num_key_value_heads=4,
                       vocab_size=32002, v_layers=2, v_intermediate=288, v_image=378, num_image_tokens=16, tokenizer_model_max_length=1024,
                       image_start_id=32000)
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(3, 31999, (2, 588), generator=g)
    ids[:, 0] = 1
    ids[:, 21], ids[:, 22], ids[:, 23] = 32000, -200, 32001
    ids[1, 300:] = 0
    labels = torch.full_like(ids, -100)
    labels[0, 200:] = ids[0, 200:]
    labels[1, 18:300] = ids[1, 18:300]
    labels[1, 22] = -200
    mask = torch.ones_like(ids, dtype=torch.bool)
    mask[1, 300:] = False
    images = torch.randn(2, 3, 378, 378, generator=g)
    sd = init_state_dict(cfg, seed=12, fast_big=True)
    kw = dict(probe_layers=(2,), grad_layers=(0, 3), embed_grad=True)
    host = full_depth(lambda k: sd[k].clone(), cfg, ids, mask, labels, images, **kw)
    dev = full_depth(lambda k: sd[k].clone(), cfg, ids, mask, labels, images, device=DEV, **kw)
    assert torch.equal(host["labels"], dev["labels"]) and torch.equal(host["image_positions"], dev["image_positions"]) and host["n_rows"] == dev["n_rows"]
    assert abs(host["loss"] - dev["loss"]) <= 2e-6 * abs(host["loss"]) and abs(host["loss_image_ar"] - dev["loss_image_ar"]) <= 2e-6
    va = host["attention_mask"]
    assert rel(dev["hidden_states"][va], host["hidden_states"][va]) < 1e-5 and rel(dev["probes"][2][va], host["probes"][2][va]) < 1e-5
    assert rel(dev["raw_hidden"], host["raw_hidden"]) < 1e-5
    worst = max((rel(dev["grads"][k], v), k) for k, v in host["grads"].items())
    print(f"\n   device vs host evaluation of the fp32 oracle: worst gradient rel diff {worst[0]:.2e} ({worst[1]}), "
          f"host {host['seconds']['total']:.1f}s device {dev['seconds']['total']:.1f}s")
    assert set(dev["grads"]) == set(host["grads"]) and len(host["grads"]) == 18 + 2 + 4 + 4 + 1 and worst[0] < 2e-4


# ------------------------------------------------------------------ depth beyond two layers for configs[0] and configs[2] (round 4)
def _depth_check(cfg, ids, labels, mask, images, seed, what, sd_hook=None, hidden_factor=1.15, key=None):
    """HIP model (bf16) vs the LAYER-STREAMED fp32 oracle (oracle/ref_stream.py, pinned to the plain oracle by tests/test_oracle_stream.py)
    on the same bf16-rounded weights, with the SAME streamed oracle run in bf16 -- the reference stack's own arithmetic -- as the yardstick
    for what depth does to bf16: loss at 1e-3 (north_star), final hidden rows no farther from the fp32 truth than 1.15 x the bf16
    oracle, every gradient of the first and the last decoder layer, the final norm, lm_head, vision_head and mm_projector within
    max(1.5 x the bf16 oracle's own distance, 3.3e-2)."""
    from oracle.ref_stream import full_depth
    sd16 = init_state_dict(cfg, seed=seed, dtype=torch.bfloat16, fast_big=True)
    if sd_hook is not None:
        sd_hook(sd16)
    model = hip_model(cfg, sd16)
    model.train()
    out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), images=images.to(DEV).bfloat16())
    plan = model.prepare_inputs_labels_for_multimodal(ids.to(DEV), None, mask.to(DEV), None, labels.to(DEV), images.to(DEV).bfloat16())
    out.loss.backward()
    torch.cuda.synchronize()
    NL = cfg.num_hidden_layers
    img16 = images.bfloat16()
    t0 = time.time()
    ref = oracle_fp32(lambda k: sd16[k].float(), lambda k: sd16[k].to(DEV).float(), cfg, ids, mask, labels, img16.float(), grad_layers=(0, NL - 1))
    t1 = time.time()
    yard = load_yardstick(key) if key else None
    from_fixture = yard is not None
    if yard is None:
        ref16 = full_depth(lambda k: sd16[k], cfg, ids, mask, labels, img16, grad_layers=(0, NL - 1), backward=True)
        va = ref["attention_mask"]
        yard = dict(oracle_fp32_loss=ref["loss"], loss=ref16["loss"], hidden_final=rel(ref16["hidden_states"].float()[va], ref["hidden_states"][va]),
                    grads={k: rel(ref16["grads"][k].float(), g) for k, g in ref["grads"].items()}, rows=ref["n_rows"])
        if RECORD_YARDSTICK and key:
            record_yardstick(key, yard)
    else:
        # the recorded yardstick belongs to THESE weights and inputs: the fp32 oracle must reproduce the loss it was recorded beside
        assert abs(ref["loss"] - yard["oracle_fp32_loss"]) <= 2e-5 * abs(ref["loss"]) and ref["n_rows"] == yard["rows"], (
            f"{key}: the fp32 oracle gives loss {ref['loss']} on rows {ref['n_rows']}, the recorded yardstick was taken beside "
            f"{yard['oracle_fp32_loss']} / {yard['rows']} -- re-record with MM355_RECORD_YARDSTICK=1")
    print(f"\n   {what}: streamed oracle fp32 ({'host' if ORACLE_FP32_DEVICE == 'cpu' or RECORD_YARDSTICK else ORACLE_FP32_DEVICE}) {t1 - t0:.0f}s, "
          f"bf16 yardstick {'from the recorded fixture' if from_fixture else f'{time.time() - t1:.0f}s on the host'}; rows per sample {ref['n_rows']}")
    assert torch.equal(plan[5].cpu(), ref["labels"]) and torch.equal(plan[6].cpu(), ref["image_positions"])
    assert torch.equal(plan[2].cpu().bool(), ref["attention_mask"])
    got, want = float(out.loss.detach()), ref["loss"]
    print(f"   loss hip={got:.5f} oracle-fp32={want:.5f} oracle-bf16={yard['loss']:.5f}  lang {model.loss_language:.5f}/{ref['loss_language']:.5f}  "
          f"img {model.loss_image_ar:.5f}/{ref['loss_image_ar']:.5f}")
    assert abs(got - want) <= 1e-3 * abs(want), (got, want)
    assert abs(model.loss_language - ref["loss_language"]) <= 1e-3 * abs(want)
    assert abs(model.loss_image_ar - ref["loss_image_ar"]) <= 1e-3
    valid = ref["attention_mask"]
    e_h = rel(out.hidden_states.float().cpu()[valid], ref["hidden_states"][valid])
    e_16 = yard["hidden_final"]
    print(f"   final hidden rows after {NL} layers vs fp32: hip {e_h:.3e}  oracle-bf16 {e_16:.3e}")
    assert e_h <= max(hidden_factor * e_16, 8e-3), (e_h, e_16)
    params = dict(model.named_parameters())
    n, worst = 0, (0.0, "", 0.0)
    for k, g in ref["grads"].items():
        assert params[k].grad is not None, k
        e, e16 = rel(params[k].grad, g), yard["grads"][k]
        worst = max(worst, (e, k, e16))
        assert e <= max(1.5 * e16, 3.3e-2), (k, e, e16)
        n += 1
    print(f"   {n} gradient tensors (layers 0 and {NL - 1}, heads, projector), worst rel err vs fp32: hip {worst[0]:.3e} (oracle-bf16 {worst[2]:.3e}) {worst[1]}")
    return n


def test_sharp_attention_logits_of_30_against_streamed_oracle():
    """A model whose attention is SHARP: the q / k projections of a 2-layer decoder (4 query heads of 128 on one KV head: the default
    d = 128 streams) scaled by 8 and 5, i.e. attention logits 40 x those of the N(0, 0.02) initialisation every other model test runs with
    (near-uniform attention): standard deviation ~ 8, extremes beyond +-30 -- the regime of a trained checkpoint's sinks and retrieval
    heads, where the forward stream's deferred-rescale branch fires and a pre-rounded operand would show.  Same yardsticks as the depth
    tests: loss at 1e-3, hidden rows within 1.3 x the bf16 oracle's own distance from fp32, every gradient within max(1.5 x, 3.3e-2)."""
    cfg = OracleConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=1,
                       vocab_size=32002, v_layers=2, v_intermediate=144, v_image=56, num_image_tokens=16, tokenizer_model_max_length=1024,
                       image_start_id=32000)
    g = torch.Generator().manual_seed(77)
    T_ = 700
    ids = torch.randint(3, 31999, (2, T_), generator=g)
    ids[:, 0] = 1
    ids[:, 21], ids[:, 22], ids[:, 23] = 32000, -200, 32001
    ids[1, 400:] = 0
    labels = torch.full_like(ids, -100)
    labels[0, 300:] = ids[0, 300:]
    labels[1, 18:400] = ids[1, 18:400]
    labels[1, 22] = -200
    mask = torch.ones_like(ids, dtype=torch.bool)
    mask[1, 400:] = False
    images = torch.randn(2, 3, 56, 56, generator=g)

    def sharpen(sd):
        for k in sd:
            if k.endswith("self_attn.q_proj.weight"):
                sd[k] = (sd[k].float() * 8.0).to(sd[k].dtype)
            elif k.endswith("self_attn.k_proj.weight"):
                sd[k] = (sd[k].float() * 5.0).to(sd[k].dtype)

    # how sharp: layer-0 logits of the oracle on these weights (printed, and required to reach the regime the test is named after)
    sd = init_state_dict(cfg, seed=5, dtype=torch.bfloat16, fast_big=True)
    sharpen(sd)
    x = sd["model.embed_tokens.weight"].float()[ids[0, 24:]]
    x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg.rms_norm_eps) * sd["model.layers.0.input_layernorm.weight"].float()
    qh = (x @ sd["model.layers.0.self_attn.q_proj.weight"].float().t()).view(-1, 4, 128)
    kh = (x @ sd["model.layers.0.self_attn.k_proj.weight"].float().t()).view(-1, 1, 128)
    s0 = torch.einsum("lhd,md->hlm", qh, kh[:, 0]) * 128 ** -0.5
    print(f"\n   layer-0 attention logits (before RoPE): std {float(s0.std()):.1f}, extremes {float(s0.min()):.1f} .. {float(s0.max()):.1f}")
    assert float(s0.abs().max()) >= 30.0
    # (sharp softmaxes amplify every bf16 rounding upstream of them: the bf16 oracle itself sits 5 % (hidden) / 5 - 20 % (gradients) from fp32
    # here -- measured on the CPU -- and two equally good roundings differ by chance, hence 1.3 x instead of the depth tests' 1.15 x)
    assert _depth_check(cfg, ids, labels, mask, images, seed=5, what="sharp attention (logits x 40)", sd_hook=sharpen, hidden_factor=1.3,
                        key="sharp_attention_logits_30") >= 20


def test_configs0_tinyllama_full_depth_22_layers_against_streamed_oracle():
    """BASELINE configs[0] at its REAL depth (round 3 ran 2 of the 22 decoder layers): TinyLlama-1.1B geometry (22 layers, h 2048,
    32 query / 4 KV heads of size 64 -> the generic attention kernels, I 5632, V 32002, <image_start> = 32000) + the SO400M/14-384 tower
    at its real 27 layers, 1 prompt image (256 tokens) + 128 text ids, plus a generation sample (answer-side image)."""
    cfg = OracleConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32, num_key_value_heads=4,
                       vocab_size=32002, rope_theta=10000.0, v_layers=27, num_image_tokens=256, tokenizer_model_max_length=2048,
                       image_start_id=32000)
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(3, 31999, (2, 129), generator=g)
    ids[:, 0] = 1
    ids[:, 21], ids[:, 22], ids[:, 23] = 32000, -200, 32001
    labels = torch.full_like(ids, -100)
    labels[0, -64:] = ids[0, -64:]
    labels[1, 18:] = ids[1, 18:]
    labels[1, 22] = -200
    mask = torch.ones_like(ids, dtype=torch.bool)
    images = torch.randn(2, 3, 384, 384, generator=g)
    assert _depth_check(cfg, ids, labels, mask, images, seed=33, what="configs[0] TinyLlama-1.1B, 22 + 27 layers", key="configs0_22_layers") == 28


def test_configs2_shape_8_frames_seq4096_eight_layers_against_streamed_oracle():
    """BASELINE configs[2]'s shape beyond two layers: LLaMA-3-8B widths, spliced length exactly 4096 with EIGHT prompt-side frames of 256
    tokens, EIGHT decoder layers (+ two tower layers), and a shorter second sample with an answer-side frame after two prompt frames."""
    cfg = OracleConfig(num_hidden_layers=8, v_layers=2, num_image_tokens=256, tokenizer_model_max_length=4096)
    g = torch.Generator().manual_seed(4321)
    L, T_img = 4096, 256
    n_ids = L - 8 * (T_img - 1)
    ids = torch.full((2, n_ids), 128001, dtype=torch.long)
    row = torch.randint(0, 127999, (n_ids,), generator=g)
    row[0] = row[1] = 128000
    for f in range(8):
        p = 22 + 3 * f
        row[p], row[p + 1], row[p + 2] = 128256, -200, 128257
    ids[0] = row
    lab0 = torch.full((n_ids,), -100, dtype=torch.long)
    lab0[-512:] = row[-512:]
    short = 604
    r1 = torch.randint(0, 127999, (short,), generator=g)
    r1[0] = r1[1] = 128000
    for p in (10, 13):
        r1[p], r1[p + 1], r1[p + 2] = 128256, -200, 128257
    r1[600], r1[601], r1[602], r1[603] = 128256, -200, 128257, 128009
    ids[1, :short] = r1
    lab1 = torch.full((n_ids,), -100, dtype=torch.long)
    lab1[300:604] = r1[300:604]
    lab1[601] = -200
    labels = torch.stack([lab0, lab1])
    mask = ids.ne(128001)
    images = torch.randn(8 + 3, 3, 384, 384, generator=g)
    assert _depth_check(cfg, ids, labels, mask, images, seed=78, what="configs[2] shape (8 frames, L = 4096), 8 layers", key="configs2_8_layers") == 28


# ------------------------------------------------------------------ BASELINE configs[3]: generation-mode finetune (all samples regress an image)
def test_configs3_all_generation_8b_widths_against_oracle():
    """BASELINE configs[3] (text -> visual-embedding regression, MetaCLIP-style pairs): every sample is
    [BOS, BOS, prompt text, assistant text, <image_start>, <image>, <image_end>, <|eot_id|>] with the labels live from the assistant
    part on, so R = B * 256 regression rows feed vision_head + the cosine loss and NO prompt-side image exists; LLaMA-3-8B widths,
    two decoder and two tower layers, three samples of different lengths (ragged right padding)."""
    cfg = OracleConfig(num_hidden_layers=2, v_layers=2, num_image_tokens=256, tokenizer_model_max_length=4096)
    g = torch.Generator().manual_seed(3303)
    B, n_ids = 3, 72
    ids = torch.full((B, n_ids), 128001, dtype=torch.long)
    labels = torch.full((B, n_ids), -100, dtype=torch.long)
    for b, (n_prompt, n_answer) in enumerate(((32, 16), (20, 8), (40, 20))):
        n = 2 + n_prompt + n_answer + 4
        row = torch.randint(0, 127999, (n,), generator=g)
        row[0] = row[1] = 128000
        row[-4], row[-3], row[-2], row[-1] = 128256, -200, 128257, 128009
        ids[b, :n] = row
        labels[b, 2 + n_prompt:n] = row[2 + n_prompt:]
        labels[b, n - 3] = -200
    mask = ids.ne(128001)
    images = torch.randn(B, 3, 384, 384, generator=g)
    n = _fullwidth_check(cfg, ids, labels, mask, images, seed=53, grad_tol=C3_GRAD_TOL, hidden_tol=1.85e-2,
                         what="configs[3] all-generation (3 x 256 regression rows)")
    assert n >= 20


C3_GRAD_TOL = 4.6e-2                                       # 1.5 x measured (3.07e-2, layers.1.q_proj); hidden 1.23e-2 -> 1.85e-2


# ------------------------------------------------------------------ BASELINE configs[4] shape: LLaMA-3-70B widths under ZeRO-3 + recompute
C4_GRAD_TOL, C4_GRAD_TOL_QK = 9.4e-2, 9.6e-2                  # 1.5 x measured at L = 4096: 6.27e-2 (layers.1.input_layernorm), q / k 6.40e-2


def test_configs4_shape_70b_widths_zero3_recompute_against_oracle():
    """BASELINE configs[4] (LLaMA-3-70B + SigLIP-SO400M, seq 4096, mixed understanding + generation batch, ZeRO-3): the 70B LAYER
    geometry (h 8192, 64 query / 8 KV heads of 128 -- eight query heads per KV group --, I 28672, V 128258) with two decoder and two tower
    layers, run the way the 70B recipe runs: decoder-layer parameters sharded (Zero3AdamW hooks: gathered per layer in forward, recompute
    and backward, gradients leaving through the rotating slots) with `gradient_checkpointing`.  One understanding sample of the recipe's
    full 4096 spliced tokens (2 prompt frames) and one shorter generation sample; loss, valid hidden rows and every gradient against the
    fp32 oracle (oracle/ref_stream.py: per sample on its valid rows, layer by layer -- the padded autograd form needs > 100 GB at
    this size)."""
    from oracle.ref_stream import full_depth
    from metamorph_amd import functional as F
    from metamorph_amd.zero2 import tag_segments
    from metamorph_amd.zero3 import Zero3AdamW
    cfg = OracleConfig(hidden_size=8192, intermediate_size=28672, num_attention_heads=64, num_key_value_heads=8, num_hidden_layers=2,
                       v_layers=1, num_image_tokens=256, tokenizer_model_max_length=4096)
    g = torch.Generator().manual_seed(7042)
    L, T_img = 4096, 256                                         # the recipe's sequence length
    n_ids = L - 2 * (T_img - 1)
    ids = torch.full((2, n_ids), 128001, dtype=torch.long)
    row = torch.randint(0, 127999, (n_ids,), generator=g)
    row[0] = row[1] = 128000
    for p in (22, 25):
        row[p], row[p + 1], row[p + 2] = 128256, -200, 128257
    ids[0] = row
    lab0 = torch.full((n_ids,), -100, dtype=torch.long)
    lab0[-256:] = row[-256:]
    short = 404                                                  # generation sample: text, an answer-side image, eot; then padding
    r1 = torch.randint(0, 127999, (short,), generator=g)
    r1[0] = r1[1] = 128000
    r1[400], r1[401], r1[402], r1[403] = 128256, -200, 128257, 128009
    ids[1, :short] = r1
    lab1 = torch.full((n_ids,), -100, dtype=torch.long)
    lab1[200:404] = r1[200:404]
    lab1[401] = -200
    labels = torch.stack([lab0, lab1])
    mask = ids.ne(128001)
    images = torch.randn(3, 3, 384, 384, generator=g)

    sd16 = init_state_dict(cfg, seed=91, dtype=torch.bfloat16, fast_big=True)
    model = hip_model(cfg, sd16)
    model.train()
    model.gradient_checkpointing_enable()
    tag_segments(model)
    params = [p for p in model.parameters() if p.requires_grad]
    names = {id(p): n for n, p in model.named_parameters()}
    opt = Zero3AdamW(params, lr=1e-5, max_grad_norm=1.0, param_slots=2, grad_slots=1).enable_hooks()
    try:
        assert all(p.data.numel() == 0 for l in model.get_model().layers for p in l.parameters())      # sharded: no resident layer weights
        out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), images=images.to(DEV).bfloat16())
        out.loss.backward()
        opt.synchronize()
        opt._drain_grad_slots()
        # the oracle, fp32, on the same bf16-rounded weights
        ref = oracle_fp32(lambda k: sd16[k].float(), lambda k: sd16[k].to(DEV).float(), cfg, ids, mask, labels, images.bfloat16().float(),
                          grad_layers=(0, 1), embed_grad=True)
        got, want = float(out.loss.detach()), ref["loss"]
        print(f"\n   configs[4] shape (L = {L}, rows {ref['n_rows']}): loss hip={got:.5f} oracle-fp32={want:.5f} img={model.loss_image_ar:.5f}/"
              f"{ref['loss_image_ar']:.5f}  oracle {ref['seconds']['total']:.0f}s")
        assert abs(got - want) <= 1e-3 * abs(want)
        assert abs(model.loss_image_ar - ref["loss_image_ar"]) <= 1e-3
        valid = ref["attention_mask"]
        e = rel(out.hidden_states.float().cpu()[valid], ref["hidden_states"].detach()[valid])
        print(f"   hidden rel err {e:.3e}")
        assert e <= 3.3e-2                                        # measured 2.2e-2 at h = 8192 / I = 28672, L = 4096
        # gradients: resident tensors in p.grad / the flat resident buffers, sharded layers in the gradient shards (world 1: whole segment)
        n, worst, worst_rest = 0, (0.0, ""), (0.0, "")
        for sg in opt.segs:
            for p, o, shp in zip(sg["params"], sg["offs"], sg["shapes"]):
                numel = 1
                for dd in shp:
                    numel *= dd
                src = sg["g_shard"] if sg["sharded"] else sg["grad"]
                gr = src[o:o + numel].view(shp)
                if not sg["sharded"] and p.grad is not None:      # autograd-routed gradients (final norm) reach the flat buffer at step()
                    gr = p.grad
                name = names[id(p)]
                e = rel(gr, ref["grads"][name])
                worst = max(worst, (e, name))
                qk = "q_proj" in name or "k_proj" in name
                worst_rest = max(worst_rest, (0.0, "") if qk else (e, name))
                n += 1
        print(f"   {n} gradient tensors (ZeRO-3 shards + resident), worst rel err {worst[0]:.3e} ({worst[1]}); outside q/k {worst_rest[0]:.3e} ({worst_rest[1]})")
        assert n >= 20
        # q / k projections of a random-weight model receive near-noise gradients (scores ~ uniform)
        assert worst[0] <= C4_GRAD_TOL_QK and worst_rest[0] <= C4_GRAD_TOL, (worst, worst_rest)
    finally:
        F.set_layer_grad_hook(None)
        F.set_param_ready_hook(None)
