"""A context-free word tokenizer used by the A1/A2 golden vectors (TEST INFRASTRUCTURE).

No tokenizer files can be downloaded in the build container, so the reference's
`tokenizer_image_token` (metamorph/mm_utils.py:191-214) is pinned with this stand-in: it has the
two properties that function relies on -- `tokenizer(text).input_ids` and `.bos_token_id` -- and,
like the Llama-3 tokenizer, (a) prepends BOS to every call when `add_bos` is set and (b) maps the
literal string "<|begin_of_text|>" to the BOS id, which is what produces the double BOS the
reference emits with its llama3 template.
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace

SPECIALS = {
    "<|begin_of_text|>": 128000,
    "<|end_of_text|>": 128001,
    "<|start_header_id|>": 128006,
    "<|end_header_id|>": 128007,
    "<|eot_id|>": 128009,
    "<image_start>": 128256,
    "<image_end>": 128257,
}


class FakeTokenizer:
    def __init__(self, add_bos=True, bos_token_id=128000, pad_token_id=128001, model_max_length=4096):
        self.add_bos = add_bos
        self.bos_token_id = bos_token_id
        self.pad_token_id = pad_token_id
        self.model_max_length = model_max_length

    def _word(self, w):
        return 3 + zlib.crc32(w.encode("utf-8")) % 127000

    def encode_plain(self, text):
        ids = []
        # split so that specials are their own tokens even when glued to words
        for sp in SPECIALS:
            text = text.replace(sp, f" {sp} ")
        for w in text.split():
            ids.append(SPECIALS[w] if w in SPECIALS else self._word(w))
        return ids

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=False):
        if isinstance(text, (list, tuple)):                  # batch form used by the reference's text-only preprocessing
            import torch
            rows = []
            for t in text:
                ids = ([self.bos_token_id] if self.add_bos else []) + self.encode_plain(t)
                if truncation and max_length is not None:
                    ids = ids[:max_length]
                rows.append(ids)
            n = max(len(r) for r in rows)
            rows = [r + [self.pad_token_id] * (n - len(r)) for r in rows]
            return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long) if return_tensors == "pt" else rows)
        ids = self.encode_plain(text)
        if self.add_bos:
            ids = [self.bos_token_id] + ids
        return SimpleNamespace(input_ids=ids)


class VocabTokenizer(FakeTokenizer):
    """FakeTokenizer + the three extra things the rest of the reference's Python surface asks a tokenizer for: `batch_decode`
    (KeywordsStoppingCriteria, mm_utils.py:226-258), `add_tokens` / `__len__` (initialize_vision_tokenizer, metamorph_arch.py:427-469).
    Words come from a fixed list so ids decode back; unknown words share one id."""

    WORDS = ("the cat sat on a mat ### stop </s> Human : Assistant answer is yes no maybe done . , ! image of red blue green "
             "one two three four five").split()

    def __init__(self, base_vocab=512, **kw):
        super().__init__(**kw)
        self.base_vocab = base_vocab
        self.word_id = {w: 10 + i for i, w in enumerate(self.WORDS)}
        self.id_word = {i: w for w, i in self.word_id.items()}
        self.added = []

    def _word(self, w):
        if w in self.added:
            return self.base_vocab + self.added.index(w)
        return self.word_id.get(w, 9)

    def encode_plain(self, text):
        for w in self.added:
            text = text.replace(w, f" {w} ")
        return super().encode_plain(text)

    def add_tokens(self, tokens, special_tokens=False):
        new = [t for t in tokens if t not in self.added]
        self.added.extend(new)
        return len(new)

    def __len__(self):
        return self.base_vocab + len(self.added)

    def batch_decode(self, ids, skip_special_tokens=False):
        special = set(SPECIALS.values()) | {self.bos_token_id, self.pad_token_id}
        inv = {v: k for k, v in SPECIALS.items()}
        out = []
        for row in ids:
            words = []
            for t in (row.tolist() if hasattr(row, "tolist") else list(row)):
                if t in special:
                    if not skip_special_tokens:
                        words.append(inv.get(t, f"<{t}>"))
                elif t >= self.base_vocab and t - self.base_vocab < len(self.added):
                    words.append(self.added[t - self.base_vocab])
                else:
                    words.append(self.id_word.get(t, "<unk>"))
            out.append(" ".join(words))
        return out
