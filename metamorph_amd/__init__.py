"""metamorph_amd -- MI355X (gfx950) native implementation of the MetaMorph interleaved vision-text
forward/backward hot path, behind the reference's `metamorph.model` / `metamorph.mm_utils` API.

    from metamorph_amd.model import MetaMorphLlamaForCausalLM      # drop-in for metamorph.model
    from metamorph_amd.mm_utils import tokenizer_image_token

All arithmetic is hand-written HIP in libmm355.so (C ABI: include/mm355.h); there is no eager fallback.
"""
__version__ = "0.1.0"
