"""Reference metamorph/model/multimodal_encoder/builder.py:11-14."""
from .siglip_encoder import SiglipVisionTower


def build_vision_tower(vision_tower_cfg, **kwargs):
    vision_tower = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    return SiglipVisionTower(vision_tower, args=vision_tower_cfg, **kwargs)
