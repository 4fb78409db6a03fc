"""Convenience constructor used by bench.py, smoke() and the tests: a MetaMorphLlamaForCausalLM with a
(randomly initialised or state-dict supplied) SigLIP tower, without touching the network."""
from __future__ import annotations

import torch

from .model import MetaMorphConfig, MetaMorphLlamaForCausalLM

LLAMA3_8B = dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                 num_key_value_heads=8, vocab_size=128258, rms_norm_eps=1e-5, rope_theta=500000.0)
TINYLLAMA_1B = dict(hidden_size=2048, intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32,
                    num_key_value_heads=4, vocab_size=32002, rms_norm_eps=1e-5, rope_theta=10000.0)


def build_model(llm: dict, vision_geometry: dict | None = None, *, num_image_tokens=256, mm_projector_type="mlp2x_gelu",
                vision_head="mlp", normalize_vision=True, apply_softmax=False, use_vision_ar=True, vision_coef=1.0, max_length=4096,
                padding_side="right", image_start_id=None, image_token_reduction="interpolation", state_dict=None, device=None,
                dtype=torch.bfloat16, init_on_device=False):
    llm = dict(llm)
    llm.setdefault("max_position_embeddings", 8192)
    llm.setdefault("tie_word_embeddings", False)
    cfg = MetaMorphConfig(attention_bias=False, **llm)
    cfg.mm_vision_tower = "siglip/CLIP-ViT-SO400M-14-384"
    cfg.mm_projector_type = mm_projector_type
    cfg.mm_hidden_size = (vision_geometry or {}).get("hidden_size", 1152) * (4 if image_token_reduction == "concat_interpolation" else 1)
    cfg.num_image_tokens = num_image_tokens
    cfg.image_token_reduction = image_token_reduction
    cfg.freeze_vision = True
    cfg.normalize_vision = normalize_vision
    cfg.apply_softmax = apply_softmax
    cfg.mm_vision_select_layer = -1
    cfg.mm_vision_geometry = vision_geometry
    cfg.tokenizer_model_max_length = max_length
    cfg.tokenizer_padding_side = padding_side
    cfg.vision_head_type = vision_head
    if image_start_id is not None:
        cfg.image_start_id = image_start_id
    ctx = torch.device(device) if (init_on_device and device is not None) else torch.device("cpu")
    with ctx:
        model = MetaMorphLlamaForCausalLM(cfg, use_vision_ar=use_vision_ar, vision_head=vision_head, vision_coef=vision_coef,
                                          normalize_vision=normalize_vision, apply_softmax=apply_softmax, vision_delay_load=True)
        tower_sd = None
        if state_dict is not None:                           # mixers of the 'mlpmixer' reduction live on the tower wrapper
            tower_sd = {k[len("model.vision_tower."):]: v for k, v in state_dict.items() if k.startswith("model.vision_tower.")
                        and not k.startswith("model.vision_tower.vision_tower.")} or None
        model.get_model().vision_tower.load_model(random_init=True, state_dict=tower_sd)
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        missing = [k for k in missing if "post_layernorm" not in k]
        if missing or unexpected:
            raise RuntimeError(f"state dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
    model.to(dtype=dtype)
    if device is not None:
        model.to(device)
    for n, p in model.named_parameters():
        if "vision_tower" in n or "vision_proj" in n:
            p.requires_grad_(False)
    return model
