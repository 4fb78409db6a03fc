"""Host-side helpers with the reference's names and semantics (reference metamorph/mm_utils.py).

Pure integer / PIL work that runs on the CPU in the reference as well; `tokenizer_image_token` is a
bit-exact contract pinned by tests/golden/a1_tokenizer.json.
"""
from __future__ import annotations

import torch

from .constants import IMAGE_TOKEN_INDEX


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise `prompt` around literal '<image>' markers (reference mm_utils.py:191-214).

    Every chunk between markers is tokenised on its own.  If the first chunk starts with BOS, that single
    BOS is kept and the first token of every chunk (the tokenizer's automatic BOS) is dropped.  Each marker
    becomes one `image_token_index`.
    """
    chunks = [tokenizer(piece).input_ids for piece in prompt.split("<image>")]
    strip = 1 if (chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id) else 0
    ids = [chunks[0][0]] if strip else []
    last = len(chunks) - 1
    for i, piece in enumerate(chunks):
        ids.extend(piece[strip:])
        if i != last:
            ids.append(image_token_index)
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def expand2square(pil_img, background_color):
    """Pad a PIL image to a square with `background_color`, centred (reference mm_utils.py:158-170)."""
    from PIL import Image
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def process_images(images, image_processor, model_cfg):
    """'pad' -> expand2square + processor per image; default -> processor on the list
    (reference mm_utils.py:172-188; the 'anyres' branch is dead in every shipped recipe and is not provided)."""
    mode = getattr(model_cfg, "image_aspect_ratio", None)
    if mode == "pad":
        out = []
        for im in images:
            im = expand2square(im, tuple(int(x * 255) for x in image_processor.image_mean))
            out.append(image_processor.preprocess(im, return_tensors="pt")["pixel_values"][0])
        if all(x.shape == out[0].shape for x in out):
            out = torch.stack(out, dim=0)
        return out
    if mode == "anyres":
        raise NotImplementedError("image_aspect_ratio='anyres' is out of scope (dead in the reference's recipes)")
    return image_processor(images, return_tensors="pt")["pixel_values"]


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria:
    """Stop when any keyword (as ids or decoded text) ends the sequence (reference mm_utils.py:226-258)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def _one(self, output_ids):
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        for kid in self.keyword_ids:
            kid = kid.to(output_ids.device)
            if torch.equal(output_ids[0, -kid.shape[0]:], kid):
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores=None, **kwargs):
        return all(self._one(output_ids[i].unsqueeze(0)) for i in range(output_ids.shape[0]))
