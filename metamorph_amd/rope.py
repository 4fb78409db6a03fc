"""RoPE parameters of a LLaMA config -> the per-band inverse frequencies the table kernel consumes.

The reference never computes these itself: `MetaMorphConfig(LlamaConfig)` (reference metamorph_llama.py:129-133) carries whatever
`rope_theta` / `rope_scaling` the base checkpoint's config.json holds -- LLaMA-3.1 8B, the README's recipe (README.md:178,187), ships
`rope_scaling = {rope_type: "llama3", factor 8, low_freq_factor 1, high_freq_factor 4, original_max_position_embeddings 8192}` -- and HF's
`LlamaRotaryEmbedding` (reached at :349-359) turns them into `inv_freq[d/2]` ONCE at construction (`ROPE_INIT_FUNCTIONS[rope_type]`), then
evaluates cos / sin(position * inv_freq) * attention_scaling in fp32 and casts to the model dtype.  Here the same one-time host step
produces `inv_freq` (fp32, torch CPU arithmetic so the bits are HF's), and `mm355_rope_table_freq` builds the bf16 tables every kernel
of the path reads (q|k|v GEMM epilogue, attention-backward epilogues, prompt pass, decode step).

Supported: "default", "llama3", "linear" (position-independent frequency tables).  "dynamic", "yarn", "longrope" and anything else are
refused by name: the first and the last change inv_freq with the sequence length at run time, and none has a reference-recorded fixture.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

SUPPORTED_ROPE_TYPES = ("default", "llama3", "linear")


@dataclass(frozen=True)
class RopeParams:
    rope_type: str
    head_dim: int
    theta: float
    inv_freq: np.ndarray            # float32 [head_dim / 2]
    attention_scaling: float

    @property
    def key(self):
        return (self.rope_type, self.head_dim, self.theta, self.inv_freq.tobytes(), self.attention_scaling)


def head_dim(config) -> int:
    """`config.head_dim` when the checkpoint states it, else hidden_size // num_attention_heads (HF LlamaAttention)."""
    d = getattr(config, "head_dim", None)
    return int(d) if d else int(config.hidden_size) // int(config.num_attention_heads)


def _rope_dict(config) -> dict:
    """One dict with 'rope_type', 'rope_theta' and the type's own fields, from either config generation: transformers >= 5 keeps
    everything in `rope_parameters`; 4.x (the reference's pin, pyproject.toml:16) has `rope_theta` + `rope_scaling` (whose type key is
    'rope_type' or, older still, 'type')."""
    rp = getattr(config, "rope_parameters", None)
    if isinstance(rp, dict) and rp:
        out = dict(rp)
    else:
        out = dict(getattr(config, "rope_scaling", None) or {})
    if "rope_theta" not in out:
        out["rope_theta"] = getattr(config, "rope_theta", None) or 10000.0
    out["rope_type"] = out.get("rope_type") or out.get("type") or "default"
    return out


def rope_params(config) -> RopeParams:
    rp = _rope_dict(config)
    kind = rp["rope_type"]
    if kind not in SUPPORTED_ROPE_TYPES:
        raise NotImplementedError(f"rope_type={kind!r}: supported RoPE variants are {SUPPORTED_ROPE_TYPES} "
                                  f"(frequency tables that do not depend on the sequence length)")
    if float(rp.get("partial_rotary_factor", 1.0)) != 1.0:
        raise NotImplementedError("partial_rotary_factor != 1: the kernels rotate whole heads")
    d = head_dim(config)
    if d % 2:
        raise ValueError(f"head_dim {d} must be even for rotate-half RoPE")
    theta = float(rp["rope_theta"])
    # fp32 torch arithmetic on the CPU, in HF's order of operations: the frequencies are the checkpoint's, bit for bit
    # (explicit device: from_pretrained constructs modules under a meta-device context)
    exponent = torch.arange(0, d, 2, dtype=torch.int64, device="cpu").to(torch.float32) / d
    inv = 1.0 / (theta ** exponent)
    if kind == "linear":
        inv = inv / float(rp["factor"])
    elif kind == "llama3":
        factor, lo, hi = float(rp["factor"]), float(rp["low_freq_factor"]), float(rp["high_freq_factor"])
        ctx = rp.get("original_max_position_embeddings") or getattr(config, "max_position_embeddings")
        wavelen = 2 * math.pi / inv
        # long wavelengths (beyond the pre-training context / lo) are stretched by `factor`, short ones (below context / hi) are kept,
        # the band between is blended
        scaled = torch.where(wavelen > ctx / lo, inv / factor, inv)
        blend = (ctx / wavelen - lo) / (hi - lo)
        mixed = (1 - blend) * scaled / factor + blend * scaled
        middle = ~(wavelen < ctx / hi) * ~(wavelen > ctx / lo)
        inv = torch.where(middle, mixed, scaled)
    return RopeParams(kind, d, theta, inv.to(torch.float32).numpy().copy(), 1.0)
