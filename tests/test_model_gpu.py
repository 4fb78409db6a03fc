"""End-to-end parity of the HIP model (metamorph_amd.model) against (a) the golden vectors recorded from the
reference itself and (b) the CPU oracle on the same seeded weights.  Needs an MI355X:  pytest -m gpu

Tolerance (north_star: "within 1e-3 bf16 tolerance"): the reference's own bf16 run differs from its fp32 run by
a data-dependent amount; we require the HIP bf16 result to be as close to the fp32 truth as the reference's bf16
result is, up to a factor, and the loss to agree with the reference-bf16 loss to 1e-3 relative (x3 for the tiny
2-layer model whose loss is ~ln(V) with ~100 target tokens).  Integer outputs are compared bit-exactly.
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle.ref_model import OracleConfig, forward as oracle_forward, init_state_dict  # noqa: E402

DEV = "cuda"


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
               vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta)
    geo = dict(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_intermediate, num_hidden_layers=cfg.v_layers,
               num_attention_heads=cfg.v_heads, image_size=cfg.v_image, patch_size=cfg.v_patch, layer_norm_eps=cfg.v_ln_eps)
    return build_model(llm, geo, num_image_tokens=cfg.num_image_tokens, use_vision_ar=cfg.use_vision_ar,
                       vision_coef=cfg.vision_coef, max_length=cfg.tokenizer_model_max_length,
                       padding_side=cfg.tokenizer_padding_side, state_dict=sd, device=DEV, **kw)


def grad_summary(t):
    f = t.detach().float().flatten().cpu()
    n = min(256, f.numel())
    idx = (torch.arange(n, dtype=torch.long) * (f.numel() - 1)) // max(n - 1, 1)
    return torch.cat([f.norm()[None], f[idx]])


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


# ------------------------------------------------------------------ A5 on the device

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "a5_*.npz"))))
def test_prepare_inputs_matches_reference(path):
    g = np.load(path)
    Timg = int(g["rows_per_image"])
    left = bool(int(g["left"]))
    cfg = tiny_cfg(hidden_size=32, intermediate_size=64, num_image_tokens=Timg, tokenizer_model_max_length=int(g["max_length"]),
                   tokenizer_padding_side="left" if left else "right")
    sd = init_state_dict(cfg, seed=11, dtype=torch.bfloat16)
    model = hip_model(cfg, sd)
    N = int(g["num_images"])
    images = torch.randn(N, 3, 56, 56, generator=torch.Generator().manual_seed(5))
    ids, lab, msk = T(g["input_ids"]).to(DEV), T(g["labels"]).to(DEV), T(g["attention_mask"]).to(DEV)
    with torch.no_grad():
        proj, feat = model.encode_images(images.to(DEV))
        out = model.prepare_inputs_labels_for_multimodal(ids, None, msk, None, lab, images.to(DEV))
    none_ids, pos_ids, att, _, emb, new_lab, img_pos, tgt = out
    assert none_ids is None and pos_ids is None
    assert torch.equal(new_lab.cpu(), T(g["out_labels"]))
    assert torch.equal(att.cpu(), T(g["out_attention_mask"]))
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
    cfg = tiny_cfg(num_image_tokens=int(g["rows_per_image"]), use_vision_ar=bool(int(g["use_vision_ar"])))
    sd = init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16)
    model = hip_model(cfg, sd)
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
        assert abs(got - ref_loss) <= 3e-3 * max(1.0, abs(ref_loss)), (got, ref_loss)
        assert abs(got - truth) <= max(3e-3 * abs(truth), 3 * abs(ref_loss - truth)), (got, truth, ref_loss)
    assert abs(model.loss_language - float(g["loss_language"])) <= 3e-3 * max(1.0, abs(ref_loss if not np.isnan(ref_loss) else 12.0))
    if np.isnan(float(g["loss_image_ar"])):
        assert np.isnan(model.loss_image_ar)          # no answer-side image rows -> NaN
    else:
        assert abs(model.loss_image_ar - float(g["loss_image_ar"])) <= 2e-2
    # hidden states: as close to fp32 truth as the reference's own bf16 run (x2) -- valid rows only
    valid = T(np.asarray(out.hidden_states.shape[:2]))  # noqa
    hs = out.hidden_states.float().cpu()
    mask = torch.zeros(hs.shape[:2], dtype=torch.bool)
    L = hs.shape[1]
    # spliced attention mask from the oracle bookkeeping
    o32 = oracle_forward(init_state_dict(cfg, seed=int(g["seed"])), cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]),
                         T(g["images"]), return_logits=False)
    mask = o32["attention_mask"]
    e_hip = rel(hs[mask], T(g32["hidden"])[mask])
    e_ref = rel(T(g["hidden"])[mask], T(g32["hidden"])[mask])
    print(f"   hidden rel err vs fp32: hip={e_hip:.4e} reference-bf16={e_ref:.4e}")
    assert e_hip <= max(2.0 * e_ref, 2e-2)
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
        worst = max(worst, e_h)
        if e_h > 3e-2:
            print(f"      {name}: rel err vs fp32 hip={e_h:.3e} reference-bf16={e_r:.3e} norm err={nerr:.3e}")
        assert e_h <= max(3.0 * e_r, 5e-2), (name, e_h, e_r)
        assert nerr <= max(5e-2, 3 * abs(float(g[k][0]) - float(g32[k][0])) / max(float(g32[k][0]), 1e-12)), (name, nerr)
        n += 1
    print(f"   {n} gradient tensors checked, worst rel err vs fp32 truth {worst:.3e}")
    assert n >= 20


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
    assert e_hip <= max(2.0 * e_ref, 2e-2)


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
        monkeypatch.setattr(F, "_DW_PAIR", paired)
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
    os.environ["MM355_DECODE_GRAPH"] = "0"
    try:
        c = model.greedy_decode(None, None, emb, max_new_tokens=8, use_cache=True)[0]
    finally:
        del os.environ["MM355_DECODE_GRAPH"]
    assert torch.equal(a, c)


# ------------------------------------------------------------------ BASELINE configs[0] geometry class (TinyLlama: d = 64, GQA 8:1)
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
    assert abs(got - want) <= 3e-3 * abs(want)
    mask = ref["attention_mask"]
    assert rel(out.hidden_states.float().cpu()[mask], ref["hidden_states"].detach()[mask]) <= 3e-2
    out.loss.backward()
    ref["loss"].backward()
    params = dict(model.named_parameters())
    n = 0
    for k, v in sd.items():
        if v.grad is None or k not in params:
            continue
        e = rel(params[k].grad, v.grad)
        assert e <= 6e-2, (k, e)
        n += 1
    assert n >= 20


# ------------------------------------------------------------------ row N4: trainable vision tower (freeze_vision=False)
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
    assert abs(got - want) <= 3e-3 * abs(want), (got, want)
    out.loss.backward()
    ref["loss"].backward()
    params = dict(model.named_parameters())
    n_tower = 0
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
            assert e <= 8e-2, (k, e)
        n_tower += 1
    print(f"\\n   trainable tower: loss hip={got:.5f} oracle={want:.5f}; {n_tower} tower gradient tensors checked")
    assert n_tower == 3 + 16 * cfg.v_layers
    # the frozen default still refuses nothing and produces no tower gradients
    model2 = hip_model(cfg, init_state_dict(cfg, seed=seed, dtype=torch.bfloat16))
    assert all(not p.requires_grad for p in model2.get_model().vision_tower.parameters())
