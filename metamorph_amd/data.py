"""Batch producer of the instruction-tuning path: llama3 conversation rendering, label masking and the
collator, with the reference's names and integer-exact behaviour (reference metamorph/train/train.py:309-332 preprocess_multimodal,
:501-597 preprocess_llama3, :1251-1284 DataCollatorForSupervisedDataset; template conversation.py:81-89, 270-278).

Pure host-side string / integer work (it runs on the CPU in the reference as well); pinned by tests/golden/n2_batch_producer.json,
recorded from the reference itself.  The output feeds `prepare_inputs_labels_for_multimodal` (row A5) unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence

import torch

from .constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IGNORE_INDEX
from .mm_utils import tokenizer_image_token

# llama3 template of the reference (conversation.py:270-278): no system text beyond BOS, header-wrapped roles, <|eot_id|> closes a turn
LLAMA3_SYSTEM = "<|begin_of_text|>"
LLAMA3_ROLES = ("<|start_header_id|>user<|end_header_id|>\n", "<|start_header_id|>assistant<|end_header_id|>\n")
LLAMA3_SEP = "<|eot_id|>"


def _legacy_offset_applies(tokenizer) -> bool:
    """The reference bumps per-round lengths by one for `legacy` sentencepiece tokenizers under tokenizers >= 0.14."""
    if not getattr(tokenizer, "legacy", False):
        return False
    try:
        import tokenizers
        from packaging import version
        return version.parse(tokenizers.__version__) >= version.parse("0.14")
    except Exception:       # pragma: no cover
        return False


def llama3_prompt(turns) -> str:
    """turns: [(role header, text or ''/None)].  A turn without text renders as the bare header (open assistant turn)."""
    out = LLAMA3_SYSTEM
    for role, text in turns:
        out += role + text + LLAMA3_SEP if text else role
    return out


def preprocess_multimodal(sources, data_args):
    """MetaMorph keeps `<image>` where the author put it; with mm_use_im_start_end every marker is wrapped in
    <image_start> ... <image_end> (in place, like the reference)."""
    if not data_args.is_multimodal:
        return sources
    marker = DEFAULT_IMAGE_TOKEN
    if data_args.mm_use_im_start_end:
        marker = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN
    for conversation in sources:
        for message in conversation:
            message["value"] = message["value"].replace(DEFAULT_IMAGE_TOKEN, marker)
    return sources


def _render(sources):
    speaker = {"human": LLAMA3_ROLES[0], "gpt": LLAMA3_ROLES[1]}
    prompts = []
    for n, conversation in enumerate(sources):
        if speaker[conversation[0]["from"]] != LLAMA3_ROLES[0]:
            conversation = conversation[1:]             # a leading assistant message is dropped
        turns = []
        for j, message in enumerate(conversation):
            role = speaker[message["from"]]
            assert role == LLAMA3_ROLES[j % 2], f"{n}"
            turns.append((role, message["value"]))
        prompts.append(llama3_prompt(turns))
    return prompts


def preprocess_llama3(sources, tokenizer, has_image: bool = False) -> Dict:
    """Conversations -> input_ids / labels with everything but the assistant answers masked (IGNORE_INDEX).

    The label walk is the reference's, quirks included: the prompt is cut at <|eot_id|>, regrouped into
    (user + assistant) rounds WITHOUT their closing <|eot_id|>, each round and its instruction part are re-tokenised on their
    own to obtain lengths, and if the accumulated length disagrees with the number of non-pad tokens (while below
    model_max_length) the whole sample is masked."""
    prompts = _render(sources)
    if has_image:
        input_ids = torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    else:
        input_ids = tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length,
                              truncation=True).input_ids
    labels = input_ids.clone()

    def n_tokens(text):
        return len(tokenizer_image_token(text, tokenizer)) if has_image else len(tokenizer(text).input_ids)

    answer_mark = LLAMA3_SEP + LLAMA3_ROLES[1]
    bump = 1 if _legacy_offset_applies(tokenizer) else 0
    for prompt, row in zip(prompts, labels):
        real = int(row.ne(tokenizer.pad_token_id).sum())
        pieces = prompt.split(LLAMA3_SEP)
        rounds = [LLAMA3_SEP.join(pieces[:2])] + [LLAMA3_SEP.join(pieces[k:k + 2]) for k in range(2, len(pieces), 2)]
        pos = 1
        row[:pos] = IGNORE_INDEX
        for k, text in enumerate(rounds):
            if text == "":
                break
            halves = text.split(answer_mark)
            if len(halves) != 2:
                break
            whole = n_tokens(text)
            instruction = n_tokens(halves[0] + answer_mark) - 1
            if k != 0:
                whole += bump
                instruction += bump
            row[pos:pos + instruction] = IGNORE_INDEX
            pos += whole
        row[pos:] = IGNORE_INDEX
        if pos < tokenizer.model_max_length and pos != real:
            row[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {pos} vs. {real}. (ignored)")
    return dict(input_ids=input_ids, labels=labels)


def preprocess(sources, tokenizer, has_image: bool = False) -> Dict:
    """The reference dispatches on the active conversation template; every shipped recipe uses llama3."""
    return preprocess_llama3(sources, tokenizer, has_image=has_image)


@dataclass
class DataCollatorForSupervisedDataset:
    """Right-pad ids with pad_token_id and labels with IGNORE_INDEX, cut to model_max_length, attention_mask = ids != pad,
    and stack the per-sample image lists into one [N, 3, H, W] tensor."""

    tokenizer: object

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        pad = self.tokenizer.pad_token_id
        limit = self.tokenizer.model_max_length
        ids = torch.nn.utils.rnn.pad_sequence([x["input_ids"] for x in instances], batch_first=True, padding_value=pad)[:, :limit]
        labels = torch.nn.utils.rnn.pad_sequence([x["labels"] for x in instances], batch_first=True, padding_value=IGNORE_INDEX)[:, :limit]
        batch = dict(input_ids=ids, labels=labels, attention_mask=ids.ne(pad))
        if "image" in instances[0]:
            batch["images"] = torch.stack([im for x in instances for im in x["image"]])
        return batch
