"""Integer bookkeeping of the hot path, restated with plain Python loops.
TEST INFRASTRUCTURE -- see oracle/__init__.py.

Rows A1 and A5 of SURVEY.md section 8a:
  * tokenizer_image_token     -- reference metamorph/mm_utils.py:191-214
  * splice bookkeeping        -- reference metamorph/model/metamorph_arch.py:245-425
Everything here is bit-exact integer work; the golden vectors in tests/golden/ pin it.
"""
from __future__ import annotations

IGNORE_INDEX = -100          # reference metamorph/constants.py:13
IMAGE_TOKEN_INDEX = -200     # reference metamorph/constants.py:14
IMAGE_START_ID = 128256      # literal at reference metamorph_arch.py:317


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX):
    """Split on the literal '<image>', tokenise every chunk, re-join with the sentinel.

    If the first chunk starts with BOS, one BOS is kept and the leading token of *every*
    chunk is dropped (mm_utils.py:199-206): the separator is emitted `offset+1` times and
    then sliced by `offset`, which leaves exactly one sentinel.
    """
    chunks = [list(tokenizer(c).input_ids) for c in prompt.split("<image>")]
    out = []
    offset = 0
    if len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        out.append(chunks[0][0])
    pieces = []
    for idx, c in enumerate(chunks):
        pieces.append(c)
        if idx != len(chunks) - 1:
            pieces.append([image_token_index] * (offset + 1))
    for p in pieces:
        out.extend(p[offset:])
    return out


def splice_bookkeeping(input_ids, labels, attention_mask, num_images_total, rows_per_image,
                       max_length, padding_side="right", image_start_id=IMAGE_START_ID):
    """Pure-loop restatement of metamorph_arch.py:245-425 (integer part only).

    input_ids / labels / attention_mask: lists of lists (B x T).  labels may be None.
    Returns a dict of plain Python lists:
      src[b][l]          token id >= 0 | ('img', image_idx, row) | None (padding row)
      labels, attention_mask, image_positions, position_ids   (B x L)
      placeholder        sorted unique image indices that are NOT regression targets
      target_keep        image indices whose features stay in target_features
    Raises IndexError like the reference when an '<image>' sentinel has no token before it
    (metamorph_arch.py:317 indexes [-1] of an empty segment).
    """
    B = len(input_ids)
    had_labels = labels is not None
    if labels is None:
        labels = [[IGNORE_INDEX] * len(r) for r in input_ids]
    if attention_mask is None:
        attention_mask = [[True] * len(r) for r in input_ids]

    rows_src, rows_lab, rows_pos = [], [], []
    placeholder = []
    img = 0
    for b in range(B):
        ids = [t for t, m in zip(input_ids[b], attention_mask[b]) if m]
        lab = [t for t, m in zip(labels[b], attention_mask[b]) if m]
        n_img = sum(1 for t in ids if t == IMAGE_TOKEN_INDEX)
        if n_img == 0:
            # text-only sample consumes one dummy image, contributing zero rows (:275-284)
            placeholder.append(img)
            img += 1
            rows_src.append(list(ids))
            rows_lab.append(list(lab))
            rows_pos.append([0] * len(ids))
            continue
        cuts = [-1] + [i for i, t in enumerate(ids) if t == IMAGE_TOKEN_INDEX] + [len(ids)]
        seg_ids = [ids[cuts[i] + 1: cuts[i + 1]] for i in range(len(cuts) - 1)]
        seg_lab = [lab[cuts[i] + 1: cuts[i + 1]] for i in range(len(cuts) - 1)]
        cur_src, cur_lab, cur_pos = [], [], []
        stop = False
        for i in range(n_img + 1):
            if not stop:
                cur_src.extend(seg_ids[i])
                cur_lab.extend(seg_lab[i])
                cur_pos.extend([0] * len(seg_ids[i]))
            if i < n_img:
                answer = seg_lab[i][-1] == image_start_id      # IndexError on empty segment
                if max_length is None:                         # :324 compares against None: `int > NoneType`
                    raise TypeError("'>' not supported between instances of 'int' and 'NoneType'")
                if len(cur_src) + rows_per_image > max_length:
                    stop = True
                    placeholder.append(img)
                else:
                    cur_src.extend(("img", img, r) for r in range(rows_per_image))
                    cur_lab.extend([IGNORE_INDEX] * rows_per_image)
                    if answer:
                        cur_pos.extend([1] * rows_per_image)
                    else:
                        placeholder.append(img)
                        cur_pos.extend([0] * rows_per_image)
                img += 1
        rows_src.append(cur_src)
        rows_lab.append(cur_lab)
        rows_pos.append(cur_pos)

    if max_length is not None:
        rows_src = [r[:max_length] for r in rows_src]
        rows_lab = [r[:max_length] for r in rows_lab]
        rows_pos = [r[:max_length] for r in rows_pos]

    L = max(len(r) for r in rows_src)
    out_src, out_lab, out_msk, out_pos, out_pid = [], [], [], [], []
    for b in range(B):
        n = len(rows_src[b])
        pad = L - n
        if padding_side == "left":
            out_src.append([None] * pad + rows_src[b])
            out_lab.append([IGNORE_INDEX] * pad + rows_lab[b])
            out_pos.append([0] * pad + rows_pos[b])
            out_msk.append([False] * pad + [True] * n)
            out_pid.append([0] * pad + list(range(n)))
        else:
            out_src.append(rows_src[b] + [None] * pad)
            out_lab.append(rows_lab[b] + [IGNORE_INDEX] * pad)
            out_pos.append(rows_pos[b] + [0] * pad)
            out_msk.append([True] * n + [False] * pad)
            out_pid.append(list(range(n)) + [0] * pad)

    ph = sorted(set(placeholder))
    keep = [i for i in range(num_images_total) if i not in set(ph)]
    return dict(src=out_src, labels=out_lab if had_labels else None, attention_mask=out_msk,
                image_positions=out_pos, position_ids=out_pid, placeholder=ph, target_keep=keep,
                images_consumed=img)
