"""Layer-streamed full-depth oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

`ref_model.forward` holds the whole state dict and lets autograd keep every layer's activations: fine for the 2-layer
fixtures, impossible for the model `bench.py` times (LLaMA-3-8B + SO400M: 32 GB of fp32 weights, > 60 GB of fp32 attention
probabilities).  This module runs THE SAME per-layer functions (`ref_model.llama_layer`, `siglip_hidden`, `reduce_features`,
`mm_projector`, `splice`, `heads` -- all pinned to the reference by tests/test_oracle_vs_golden.py) with

  * weights fetched one tensor at a time through `fetch(name) -> fp32 CPU tensor` (the caller reads them from the device model),
  * every sample processed on its own valid rows (no flops on padding; causal attention without key padding is the same
    computation as the batched form with a right-padding mask),
  * the backward pass re-running one layer at a time from the stored layer inputs (the chain d loss / d x_i is exact; only the
    layers in `grad_layers` pay for weight gradients).

Reference call sites: metamorph_llama.py:349-359 (decoder), :393-474 (heads / loss combine), siglip_encoder.py:138-213 (tower),
metamorph_arch.py:140-164, 177-425 (projector, splice).
"""
from __future__ import annotations

import time

import torch

from . import ref_model as RM
from . import ref_ops as ops

LAYER_TENSORS = ("input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                 "self_attn.o_proj.weight", "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                 "mlp.down_proj.weight")


class LazyStateDict:
    """sd[name] -> fetch(name); nothing is kept."""

    def __init__(self, fetch):
        self.fetch = fetch

    def __getitem__(self, k):
        return self.fetch(k)


def _layer_weights(fetch, i, requires_grad=False):
    p = f"model.layers.{i}."
    w = {p + k: fetch(p + k) for k in LAYER_TENSORS}
    if requires_grad:
        for v in w.values():
            v.requires_grad_(True)
    return w


def _to_cpu(o):
    if isinstance(o, torch.Tensor):
        return o.detach().cpu()
    if isinstance(o, dict):
        return {k: _to_cpu(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_to_cpu(v) for v in o)
    return o


def full_depth(fetch, cfg: RM.OracleConfig, input_ids, attention_mask, labels, images, probe_layers=(), grad_layers=(),
               head_grads=True, backward=True, log=None, embed_grad=False, device=None):
    """fp32 forward (and backward) of the whole model, layer-streamed.  Right padding only.

    device: None = where the inputs live (the host: the oracle's home, and the only place its bf16 run means "the reference stack's bf16
    arithmetic").  "cuda": the SAME functions evaluated with stock torch fp32 ops on the GPU (weights moved as they are fetched; every
    result returned on the host) -- the full-width fp32 runs of the GPU parity tests take seconds instead of minutes of host GEMMs; the
    device evaluation is itself pinned to the reference's recorded fp32 run and to the host evaluation (tests/test_model_gpu.py
    test_streamed_oracle_on_device_*)."""
    if device is not None:
        dev = torch.device(device)
        mv = lambda t: None if t is None else t.to(dev)
        with torch.device(dev):
            out = _full_depth(lambda k: fetch(k).to(dev), cfg, mv(input_ids), mv(attention_mask), mv(labels), mv(images), probe_layers,
                              grad_layers, head_grads, backward, log, embed_grad)
        return _to_cpu(out)
    return _full_depth(fetch, cfg, input_ids, attention_mask, labels, images, probe_layers, grad_layers, head_grads, backward, log, embed_grad)


def _full_depth(fetch, cfg, input_ids, attention_mask, labels, images, probe_layers, grad_layers, head_grads, backward, log, embed_grad):
    """(the body of full_depth: every tensor is created next to the inputs)

    Returns a dict: raw_hidden [N, P, hv] (tower, hidden_states[-1]), features, projected, labels / attention_mask /
    image_positions (spliced, as `ref_model.forward`), probes {n: [B, L, h] hidden rows after n decoder layers (zeros on padding)},
    hidden_states [B, L, h] (after the final norm), loss / loss_language / loss_image_ar, grads {name: tensor} for the tensors of
    `grad_layers`, and -- head_grads -- model.norm, lm_head, vision_head.*, model.mm_projector.* (embed_grad: model.embed_tokens too),
    seconds {phase: s}."""
    say = log or (lambda *_: None)
    sd = LazyStateDict(fetch)
    t_all = time.time()
    secs = {}
    t0 = time.time()
    with torch.no_grad():
        raw = RM.siglip_hidden(sd, cfg, images)
        feat = RM.reduce_features(sd, cfg, raw.to(images.dtype))
    secs["tower"] = time.time() - t0
    say(f"[oracle] tower {secs['tower']:.1f}s")
    proj_names = [f"model.mm_projector.{2 * j}.{s}" for j in range(int(cfg.mm_projector_type[3:-6])) for s in ("weight", "bias")]
    proj_w = {k: fetch(k).requires_grad_(bool(backward and head_grads)) for k in proj_names}
    proj = RM.mm_projector(proj_w, cfg, feat)
    emb = {"model.embed_tokens.weight": fetch("model.embed_tokens.weight").requires_grad_(bool(backward and embed_grad))}
    x, lab, key_valid, img_pos, target, _ = RM.splice(emb, cfg, input_ids, labels, attention_mask, proj, feat.detach().clone())
    if not (backward and embed_grad):
        del emb
    B, L, h = x.shape
    n_rows = [int(key_valid[b].sum()) for b in range(B)]
    for b in range(B):
        if not bool(key_valid[b, :n_rows[b]].all()):
            raise ValueError("full_depth handles right-padded batches")
    pos = torch.arange(L)[None]
    cos, sin = ops.rope_tables(pos, cfg.head_dim, cfg.rope_theta, x.dtype, **cfg.rope_kw)
    xs = [x[b, :n_rows[b]].detach().clone() for b in range(B)]
    inputs = []                                                  # inputs[i][b]: rows entering decoder layer i
    probes = {}
    NL = cfg.num_hidden_layers
    t0 = time.time()
    with torch.no_grad():
        for i in range(NL):
            w = _layer_weights(fetch, i)
            inputs.append(xs)
            xs = [RM.llama_layer(w, cfg, i, xs[b][None], None, cos[:, :n_rows[b]], sin[:, :n_rows[b]])[0] for b in range(B)]
            del w
            if (i + 1) in probe_layers:
                pr = torch.zeros(B, L, h)
                for b in range(B):
                    pr[b, :n_rows[b]] = xs[b]
                probes[i + 1] = pr
            say(f"[oracle] decoder layer {i + 1}/{NL} forward, {time.time() - t0:.1f}s")
    secs["decoder_forward"] = time.time() - t0

    # heads on the padded layout (padding rows are never selected by labels / image_positions)
    t0 = time.time()
    want_bwd = bool(backward)
    xl = [t.detach().requires_grad_(want_bwd) for t in xs]
    head_names = ["model.norm.weight", "lm_head.weight"] + {
        "mlp": [f"vision_head.{j}.{s}" for j in (0, 2) for s in ("weight", "bias")],
        "mlp2x_gelu": [f"vision_head.{j}.{s}" for j in (0, 2, 4) for s in ("weight", "bias")]}.get(cfg.vision_head_type,
                                                                                                ["vision_head.weight", "vision_head.bias"])
    head_w = {k: fetch(k).requires_grad_(bool(want_bwd and head_grads)) for k in head_names}
    rows = []
    for b in range(B):
        hb = ops.rmsnorm(xl[b], head_w["model.norm.weight"], cfg.rms_norm_eps)
        rows.append(torch.cat([hb, hb.new_zeros(L - n_rows[b], h)], 0) if n_rows[b] < L else hb)
    hid = torch.stack(rows, 0)
    res = RM.heads(head_w, cfg, hid, lab, img_pos, target, return_logits=False, ce_rows_only=True)
    secs["heads_forward"] = time.time() - t0
    out = {"raw_hidden": raw, "features": feat, "projected": proj.detach(), "labels": lab, "attention_mask": key_valid,
           "image_positions": img_pos, "target_features": target, "probes": probes, "hidden_states": hid.detach(),
           "loss": float(res["loss"].detach()), "loss_language": res["loss_language"], "loss_image_ar": res["loss_image_ar"],
           "n_rows": n_rows, "grads": {}, "seconds": secs}
    if not want_bwd:
        secs["total"] = time.time() - t_all
        return out

    t0 = time.time()
    res["loss"].backward()
    dx = [t.grad for t in xl]
    if head_grads:
        for k, v in head_w.items():
            out["grads"][k] = v.grad
    del head_w, hid, res, rows
    secs["heads_backward"] = time.time() - t0
    t0 = time.time()
    for i in reversed(range(NL)):
        w = _layer_weights(fetch, i, requires_grad=i in grad_layers)
        new_dx = []
        for b in range(B):
            xin = inputs[i][b].detach().requires_grad_(True)
            y = RM.llama_layer(w, cfg, i, xin[None], None, cos[:, :n_rows[b]], sin[:, :n_rows[b]])[0]
            y.backward(dx[b])
            new_dx.append(xin.grad)
            del y, xin
        dx = new_dx
        inputs[i] = None
        if i in grad_layers:
            for k, v in w.items():
                out["grads"][k] = v.grad
        del w
        say(f"[oracle] decoder layer {i + 1}/{NL} backward, {time.time() - t0:.1f}s")
    secs["decoder_backward"] = time.time() - t0
    dx0 = torch.zeros(B, L, h)
    for b in range(B):
        dx0[b, :n_rows[b]] = dx[b]
    out["d_inputs_embeds"] = dx0
    if (head_grads or embed_grad) and x.requires_grad:
        x.backward(dx0)
        if head_grads:
            for k, v in proj_w.items():
                out["grads"][k] = v.grad
        if embed_grad:                                           # dense [V, h], as torch's embedding backward produces it
            out["grads"]["model.embed_tokens.weight"] = emb["model.embed_tokens.weight"].grad
    secs["total"] = time.time() - t_all
    return out
