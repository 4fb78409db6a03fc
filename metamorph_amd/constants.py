"""Sentinel values of the reference's batch contract (reference metamorph/constants.py:13-19)."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<image_start>"
DEFAULT_IM_END_TOKEN = "<image_end>"
IMAGE_PLACEHOLDER = "<image-placeholder>"
# id of <image_start> once the two special tokens are appended to the 128256-entry Llama-3 vocabulary; the
# reference hard-codes it (metamorph_arch.py:317, metamorph_llama.py:502) -- here it is config.image_start_id.
DEFAULT_IMAGE_START_ID = 128256
DEFAULT_IMAGE_END_ID = 128257
