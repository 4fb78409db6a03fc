"""mm_projector factory with the reference's names (reference metamorph/model/multimodal_projector/builder.py:39-64)."""
from __future__ import annotations

import re

import torch.nn as nn

from ..modules import HipGELU, HipLinear, HipSoftmax


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


def build_vision_projector(config, delay_load=False, **kwargs):
    projector_type = getattr(config, "mm_projector_type", "linear")
    if projector_type == "linear":
        return HipLinear(config.mm_hidden_size, config.hidden_size)
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        depth = int(m.group(1))
        mods = [HipLinear(config.mm_hidden_size, config.hidden_size)]
        for _ in range(1, depth):
            mods.append(HipGELU())
            mods.append(HipLinear(config.hidden_size, config.hidden_size))
        return nn.Sequential(*mods)
    if projector_type == "identity":
        return IdentityMap()
    if projector_type == "mlpsoftmax":       # Linear -> Softmax(dim=-1) -> Linear (reference builder.py:45-50)
        return nn.Sequential(HipLinear(config.mm_hidden_size, config.hidden_size), HipSoftmax(dim=-1),
                             HipLinear(config.hidden_size, config.hidden_size))
    raise ValueError(f"Unknown projector type: {projector_type}")
