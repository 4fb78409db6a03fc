"""Host gather plan for the <image>/text splice.

The reference splices with a Python loop of device ops and one device sync per sample
(reference metamorph/model/metamorph_arch.py:245-425).  Here the *bookkeeping* -- which is pure integer
work and a bit-exact contract -- is done once on the host with numpy from a single copy of the [B,T]
id/label/mask arrays, and emitted as int32 index arrays that drive four HIP kernels
(mm355_splice_gather, mm355_rows_gather, mm355_embed_grad, mm355_rows_scatter_add).

Every quirk of the reference is preserved and pinned by tests/golden/a5_*.npz:
  * padding is stripped with the attention mask first (:259-260);
  * a sample without any sentinel consumes one (dummy) image and contributes none of its rows (:275-284);
  * an image is an *answer* image iff the label just before its sentinel equals <image_start> (:317)
    -- an empty segment there raises IndexError exactly like the reference;
  * an image that would make the sample longer than tokenizer_model_max_length is dropped together with
    all later text of that sample (:324-326, :304-309);
  * sequences are truncated to tokenizer_model_max_length, then padded right (or left) with zero rows,
    labels -100, image_positions 0 (:355-399);
  * regression targets keep only images that were spliced in as answer images (:415-423).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_START_ID


@dataclass
class SplicePlan:
    B: int
    L: int
    rows_per_image: int
    src: np.ndarray               # int32 [B*L]: token id >= 0 | -1 pad | -2 - feature_row
    labels: np.ndarray | None     # int64 [B,L]
    attention_mask: np.ndarray    # bool  [B,L]
    image_positions: np.ndarray   # int64 [B,L] in {0,1}
    position_ids: np.ndarray      # int64 [B,L]
    seqlens: np.ndarray           # int32 [B] number of valid rows per sample
    target_keep: np.ndarray       # int64 [Na] image indices kept as regression targets
    feat_row: np.ndarray          # int32 [N*T]: spliced row (b*L + l) of every image feature row, -1 if dropped
    pred_rows: np.ndarray         # int32 [R]: rows (b*L + t) of hidden whose NEXT position is an answer-image row
    shift_targets: np.ndarray | None   # int32 [B*L]: labels[b, t+1] (or -100) -- the CE target of row (b,t)
    ce_rows: np.ndarray | None    # int32 [n_valid] rows with a CE target (row-major)
    n_valid: int                  # number of CE targets
    emb_tok: np.ndarray           # int32 [S] unique token ids present        } segments for the
    emb_seg: np.ndarray           # int32 [S+1] segment starts into emb_pos   } embedding gradient
    emb_pos: np.ndarray           # int32 [*] spliced rows sorted by token id }
    images_consumed: int
    padding_side: str = "right"


def build_splice_plan(input_ids, labels, attention_mask, num_images: int, rows_per_image: int,
                      max_length: int | None, padding_side: str = "right",
                      image_start_id: int = DEFAULT_IMAGE_START_ID, vocab_size: int | None = None) -> SplicePlan:
    ids_all = np.asarray(input_ids)
    assert ids_all.ndim == 2
    B = ids_all.shape[0]
    had_labels = labels is not None
    lab_all = np.asarray(labels) if had_labels else np.full_like(ids_all, IGNORE_INDEX)
    msk_all = np.ones_like(ids_all, dtype=bool) if attention_mask is None else np.asarray(attention_mask).astype(bool)
    T = int(rows_per_image)

    row_src, row_lab, row_pos = [], [], []
    placeholder = []
    img = 0
    for b in range(B):
        ids = ids_all[b][msk_all[b]]
        lab = lab_all[b][msk_all[b]]
        # token ids feed an embedding gather: anything outside [0, vocab) other than the <image> sentinel raises in the reference
        # (nn.Embedding "index out of range in self", metamorph_arch.py:278,298) and must not become an out-of-bounds gather here
        bad = (ids < 0) & (ids != IMAGE_TOKEN_INDEX)
        if vocab_size is not None:
            bad |= ids >= vocab_size
        if bad.any():
            raise IndexError(f"index out of range in self: token id {int(ids[bad][0])} of sample {b} is outside [0, {vocab_size})")
        where = np.flatnonzero(ids == IMAGE_TOKEN_INDEX)
        if where.size == 0:
            placeholder.append(img)
            img += 1
            row_src.append(ids.astype(np.int64))
            row_lab.append(lab.astype(np.int64))
            row_pos.append(np.zeros(ids.shape[0], dtype=np.int64))
            continue
        cuts = np.concatenate(([-1], where, [ids.shape[0]]))
        src_parts, lab_parts, pos_parts = [], [], []
        cur_len = 0
        stopped = False
        for i in range(where.size + 1):
            seg = slice(cuts[i] + 1, cuts[i + 1])
            if not stopped:
                src_parts.append(ids[seg].astype(np.int64))
                lab_parts.append(lab[seg].astype(np.int64))
                pos_parts.append(np.zeros(cuts[i + 1] - cuts[i] - 1, dtype=np.int64))
                cur_len += cuts[i + 1] - cuts[i] - 1
            if i < where.size:
                seg_lab = lab[seg]
                if seg_lab.shape[0] == 0:
                    raise IndexError("index -1 is out of bounds for dimension 0 with size 0 "
                                     "(an <image> sentinel with no token before it; reference metamorph_arch.py:317)")
                answer = int(seg_lab[-1]) == image_start_id
                if max_length is None:
                    # a config without `tokenizer_model_max_length` only survives text-only batches in the reference: the overflow test
                    # compares against None (metamorph_arch.py:271, 324)
                    raise TypeError("'>' not supported between instances of 'int' and 'NoneType' (config.tokenizer_model_max_length is not "
                                    "set and a sample holds an <image>; reference metamorph_arch.py:324)")
                if cur_len + T > max_length:
                    stopped = True
                    placeholder.append(img)
                else:
                    src_parts.append(-2 - (img * T + np.arange(T, dtype=np.int64)))
                    lab_parts.append(np.full(T, IGNORE_INDEX, dtype=np.int64))
                    pos_parts.append(np.full(T, 1 if answer else 0, dtype=np.int64))
                    cur_len += T
                    if not answer:
                        placeholder.append(img)
                img += 1
        row_src.append(np.concatenate(src_parts))
        row_lab.append(np.concatenate(lab_parts))
        row_pos.append(np.concatenate(pos_parts))

    # the reference indexes image_features[cur_image_idx] for EVERY sentinel and every text-only sample's dummy image
    # (metamorph_arch.py:277,320): one image too few is an IndexError there, and must not become an out-of-bounds gather here
    if img > num_images:
        raise IndexError(f"index {num_images} is out of bounds for dimension 0 with size {num_images} "
                         f"(the batch consumes {img} images -- <image> sentinels plus one dummy per text-only sample -- "
                         f"but {num_images} were supplied; reference metamorph_arch.py:277,320)")
    if max_length is not None:
        row_src = [r[:max_length] for r in row_src]
        row_lab = [r[:max_length] for r in row_lab]
        row_pos = [r[:max_length] for r in row_pos]

    L = max(r.shape[0] for r in row_src)
    src = np.full((B, L), -1, dtype=np.int64)
    out_lab = np.full((B, L), IGNORE_INDEX, dtype=np.int64)
    out_pos = np.zeros((B, L), dtype=np.int64)
    out_msk = np.zeros((B, L), dtype=bool)
    out_pid = np.zeros((B, L), dtype=np.int64)
    seqlens = np.zeros(B, dtype=np.int32)
    left = padding_side == "left"
    for b in range(B):
        n = row_src[b].shape[0]
        seqlens[b] = n
        sl = slice(L - n, L) if left else slice(0, n)
        if n > 0:
            src[b, sl] = row_src[b]
            out_lab[b, sl] = row_lab[b]
            out_pos[b, sl] = row_pos[b]
            out_msk[b, sl] = True
            out_pid[b, sl] = np.arange(n)

    if num_images > 0:
        keep_mask = np.ones(num_images, dtype=bool)
        keep_mask[np.asarray(placeholder, dtype=np.int64)] = False      # IndexError if a dummy image is missing
        target_keep = np.flatnonzero(keep_mask)
    else:
        target_keep = np.zeros(0, dtype=np.int64)

    flat = src.reshape(-1)
    feat_row = np.full(max(num_images, img) * T, -1, dtype=np.int32)
    is_img = flat <= -2
    feat_row[(-2 - flat[is_img]).astype(np.int64)] = np.flatnonzero(is_img).astype(np.int32)

    nxt = np.zeros((B, L), dtype=bool)
    nxt[:, :-1] = out_pos[:, 1:] == 1
    pred_rows = np.flatnonzero(nxt.reshape(-1)).astype(np.int32)

    shift_targets = ce_rows = None
    n_valid = 0
    if had_labels:
        st = np.full((B, L), IGNORE_INDEX, dtype=np.int64)
        st[:, :-1] = out_lab[:, 1:]
        shift_targets = st.reshape(-1).astype(np.int32)
        ce_rows = np.flatnonzero(shift_targets != IGNORE_INDEX).astype(np.int32)
        n_valid = int(ce_rows.shape[0])

    tok_rows = np.flatnonzero(flat >= 0)
    order = np.argsort(flat[tok_rows], kind="stable")
    sorted_tok = flat[tok_rows][order]
    emb_pos = tok_rows[order].astype(np.int32)
    if sorted_tok.size:
        starts = np.flatnonzero(np.concatenate(([True], sorted_tok[1:] != sorted_tok[:-1])))
        emb_tok = sorted_tok[starts].astype(np.int32)
        emb_seg = np.concatenate((starts, [sorted_tok.size])).astype(np.int32)
    else:
        emb_tok = np.zeros(0, dtype=np.int32)
        emb_seg = np.zeros(1, dtype=np.int32)

    return SplicePlan(B=B, L=L, rows_per_image=T, src=flat.astype(np.int32), labels=out_lab if had_labels else None,
                      attention_mask=out_msk, image_positions=out_pos, position_ids=out_pid, seqlens=seqlens,
                      target_keep=target_keep, feat_row=feat_row, pred_rows=pred_rows, shift_targets=shift_targets,
                      ce_rows=ce_rows, n_valid=n_valid, emb_tok=emb_tok, emb_seg=emb_seg, emb_pos=emb_pos,
                      images_consumed=img, padding_side=padding_side)


def compact_row_maps(seqlens, B, L, granule=256, full=False):
    """Row maps between the right-padded layout (row b * L + l, valid for l < n_b) and the compact one (the valid rows back to back, rounded
    up to `granule` rows -- whole GEMM tiles and whole 64-row transposed vectors): (c2p int32 [rows]: padded row of every compact row, -1 in
    the tail; p2c int32 [B * L]: compact row of every padded row, -1 for padding).  full=True: c2p has B * L entries (-1 beyond the valid
    rows): any prefix of at least `rows` entries is a valid map -- the caller picks the row count (LlamaForCausalLM keeps it constant across
    steps)."""
    n = np.asarray(seqlens, dtype=np.int64)
    total = int(n.sum())
    rows = max(granule, (total + granule - 1) // granule * granule)
    c2p = np.full(max(rows, (B * L + granule - 1) // granule * granule) if full else rows, -1, dtype=np.int32)
    p2c = np.full(B * L, -1, dtype=np.int32)
    at = 0
    for b in range(B):
        k = int(n[b])
        c2p[at:at + k] = b * L + np.arange(k, dtype=np.int32)
        p2c[b * L:b * L + k] = at + np.arange(k, dtype=np.int32)
        at += k
    return c2p, p2c
