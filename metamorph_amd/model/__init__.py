"""Same public names as the reference's `metamorph.model` package."""
from .language_model.metamorph_llama import MetaMorphConfig, MetaMorphLlamaForCausalLM, MetaMorphLlamaModel
from .multimodal_encoder.builder import build_vision_tower
from .multimodal_projector.builder import build_vision_projector

__all__ = ["MetaMorphConfig", "MetaMorphLlamaForCausalLM", "MetaMorphLlamaModel", "build_vision_tower", "build_vision_projector"]
