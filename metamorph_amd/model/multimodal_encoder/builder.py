"""Vision-tower factory with the reference's name and argument meaning (reference metamorph/model/multimodal_encoder/builder.py:11-14).

The tower name comes from `mm_vision_tower` when the config carries it (a reloaded checkpoint) and from `vision_tower` otherwise
(fresh ModelArguments).  Only the SigLIP tower is on the accelerated path; the name is handed through unchanged so the tower can
resolve its geometry (and, given `weights=`, its parameters) from it.
"""
from .siglip_encoder import SiglipVisionTower

_NAME_FIELDS = ("mm_vision_tower", "vision_tower")


def _tower_name(cfg):
    for field in _NAME_FIELDS:          # the first field PRESENT wins, even when it holds None (as in the reference)
        if hasattr(cfg, field):
            return getattr(cfg, field)
    return None


def build_vision_tower(vision_tower_cfg, **kwargs):
    return SiglipVisionTower(_tower_name(vision_tower_cfg), args=vision_tower_cfg, **kwargs)
