"""MetaMorphLlamaForCausalLM on MI355X kernels -- drop-in for the reference class of the same name
(reference metamorph/model/language_model/metamorph_llama.py:129-738).

Same constructor, same `forward` signature and `CausalLMOutputWithPast` contract, same side-effect attributes
(`loss_language`, `loss_image_ar`), same state-dict keys; the arithmetic is libmm355 (hand-written gfx950
HIP) instead of transformers' LlamaModel + torch ops:

  reference                                         here
  ------------------------------------------------  --------------------------------------------------------
  HF LlamaModel, 32 x LlamaDecoderLayer             functional.DecoderLayerFn (one autograd node per layer:
    (RMSNorm, q/k/v/o Linear, RoPE, SDPA, SwiGLU)      fused qkv / gate-up GEMMs, flash attention, fused
                                                       residual epilogues, hand-written backward)
  lm_head -> fp32 logits [B,L,V] -> CrossEntropy    functional.LinearCrossEntropyFn (chunked, target rows only)
  mask-multiply + boolean index (device sync)       row gather by the host plan (no sync)
  vision_head -> F.normalize -> cosine_similarity   GEMMs + fused cosine loss kernel
  three .item() syncs per step                      device-side combine; floats materialise only when read
"""
from __future__ import annotations


import copy
import numpy as np
import torch
import torch.nn as nn
from transformers import AutoConfig, AutoModelForCausalLM, GenerationMixin, LlamaConfig, PreTrainedModel
from transformers.cache_utils import DynamicCache
from transformers.modeling_outputs import CausalLMOutputWithPast

from ... import functional as F
from ... import ops
from ...constants import DEFAULT_IMAGE_END_ID, DEFAULT_IMAGE_START_ID, IGNORE_INDEX
from ...splice_plan import SplicePlan, compact_row_maps
from ...hostmirror import host_array
from ...rope import head_dim as _head_dim, rope_params
from ..metamorph_arch import MetaMorphMetaForCausalLM, MetaMorphMetaModel, upload_plan
from ..modules import HipEmbedding, HipGELU, HipLinear, HipRMSNorm

BF16 = torch.bfloat16


class MetaMorphConfig(LlamaConfig):
    model_type = "metamorph_llama"


def _left_pad_maps(seqlens, B, L):
    """Row maps between the left-padded layout (valid rows [L - n_b, L)) and the right-padded one (valid rows [0, n_b)):
    (to_right [B*L]: source row of every right-layout row, -1 = zero row; to_left: the inverse; off [B] = L - n_b), all int32."""
    to_right = np.full(B * L, -1, dtype=np.int32)
    to_left = np.full(B * L, -1, dtype=np.int32)
    off = (L - np.asarray(seqlens, dtype=np.int64)).astype(np.int32)
    for b in range(B):
        n = int(seqlens[b])
        to_right[b * L:b * L + n] = b * L + off[b] + np.arange(n, dtype=np.int32)
        to_left[b * L + off[b]:(b + 1) * L] = b * L + np.arange(n, dtype=np.int32)
    return to_right, to_left, off


class _Attention(nn.Module):
    def __init__(self, config):
        super().__init__()
        h, d = config.hidden_size, _head_dim(config)
        self.q_proj = HipLinear(h, config.num_attention_heads * d, bias=False)
        self.k_proj = HipLinear(h, config.num_key_value_heads * d, bias=False)
        self.v_proj = HipLinear(h, config.num_key_value_heads * d, bias=False)
        self.o_proj = HipLinear(config.num_attention_heads * d, h, bias=False)


class _MLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.gate_proj = HipLinear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = HipLinear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = HipLinear(config.intermediate_size, config.hidden_size, bias=False)


class _DecoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = _Attention(config)
        self.mlp = _MLP(config)
        self.input_layernorm = HipRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = HipRMSNorm(config.hidden_size, eps=config.rms_norm_eps)


class _SeqView:
    """Read-only view of ONE sequence of a HipKVCache (introspection: tests, debugging): its left-padding and its cached length."""

    def __init__(self, cache, b):
        self._c, self._b = cache, b

    @property
    def pad(self):
        return self._c.pads[self._b]

    @property
    def length(self):
        return self._c.kv.lengths[self._b]


class HipKVCache(DynamicCache):
    """What `past_key_values` is under HF `generate()` (reference metamorph_llama.py:711-717, `use_customize_greedy=False`): a
    `transformers` Cache whose payload is ONE `functional.KVCache` for the whole batch (batch rows / beams) -- post-RoPE keys / values of
    every decoder layer in the layout the decode kernels read ([layers, batch, max_len, Hkv*d] bf16, per-row write positions on the device)
    -- plus the captured hipGraph of the per-token step, which takes all rows through the decoder in ONE pass over the weights (as the
    reference's `forward` does: the batch goes to one HF forward per step).  HF only asks a cache for its length and, under beam search,
    to re-order its batch rows (`reorder_cache`); the tensors never leave the device or change layout."""

    def __init__(self, capacity=None, **kw):
        super().__init__(**kw)
        self.capacity = capacity          # rows to allocate at the first (prompt) pass; None: prompt + 1024 + 2
        self.kv = None                    # functional.KVCache of the whole batch
        self.pads = []                    # left-padding rows of every sequence in the batch HF sees (never cached: kv.lengths count real rows)
        self.stepper = None               # functional.DecodeStepGraph
        self.meta = None

    @property
    def states(self):
        return [] if self.kv is None else [_SeqView(self, b) for b in range(len(self.pads))]

    def get_seq_length(self, layer_idx=0):
        # the length HF counts: padded prompt + generated tokens, the same for every row of a left-padded batch
        return 0 if self.kv is None else int(self.kv.lengths[0]) + self.pads[0]

    def get_max_cache_shape(self, layer_idx=0):
        return -1 if self.kv is None else int(self.kv.max_len)

    def reorder_cache(self, beam_idx):
        """Beam search: row i continues the hypothesis that lived in row beam_idx[i].  ONE gather over the batch dimension of the live
        prefix (every layer, keys and values) through a temporary, written back in place: buffers -- and the captured graph that points
        at them -- stay where they are; a row may be source and target at once."""
        idx = [int(i) for i in beam_idx.reshape(-1).tolist()]
        if len(idx) != len(self.pads):
            raise ValueError(f"reorder_cache: {len(idx)} indices for {len(self.pads)} sequences")
        if all(i == j for i, j in enumerate(idx)):
            return
        kv = self.kv
        n = max(kv.lengths)
        sel = torch.as_tensor(idx, dtype=torch.long, device=kv.k.device)
        kv.k[:, :, :n].copy_(kv.k[:, :, :n].index_select(1, sel))
        kv.v[:, :, :n].copy_(kv.v[:, :, :n].index_select(1, sel))
        kv.set_lengths([kv.lengths[j] for j in idx])
        self.pads = [self.pads[j] for j in idx]

    def crop(self, max_length):
        """HF's convention: keep the first `max_length` positions of the PADDED sequence (what get_seq_length counts); negative = drop
        the last -max_length positions.  Every row keeps max_length - pad real rows."""
        if self.kv is None:                                   # nothing cached yet (an empty DynamicCache.crop is a no-op as well)
            return
        max_length = int(max_length)
        if max_length < 0:
            max_length = self.get_seq_length() + max_length
        keep = [min(n, max(max_length - pad, 0)) for n, pad in zip(self.kv.lengths, self.pads)]
        if keep != list(self.kv.lengths):
            self.kv.set_lengths(keep)


class MetaMorphLlamaModel(MetaMorphMetaModel, nn.Module):
    """`model.*` sub-tree: embed_tokens, layers, norm, vision_tower, mm_projector, vision_proj."""
    config_class = MetaMorphConfig

    def __init__(self, config, vision_delay_load=True):
        nn.Module.__init__(self)
        self.config = config
        if getattr(config, "attention_bias", False) or getattr(config, "mlp_bias", False):
            raise NotImplementedError("attention_bias / mlp_bias are not used by LLaMA-3 and have no fused kernel")
        self.embed_tokens = HipEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([_DecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = HipRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self._init_vision(config, vision_delay_load=vision_delay_load)
        self.rope = rope_params(config)         # refuses the RoPE variants without a kernel path at construction, by name
        self._rope = None
        # set by PreTrainedModel.gradient_checkpointing_enable() (HF looks for this attribute on the sub-modules): per-layer
        # recompute in DecoderLayerFn / SiglipLayerFn instead of torch.utils.checkpoint
        self.gradient_checkpointing = False
        # MI355X extension of the checkpointing mode (288 GB of HBM rarely needs every layer recomputed): None = every decoder layer
        # (the reference's behaviour), n = only the first n decoder layers keep just their input; the rest keep their activations.
        # Recompute runs the same kernels on the same inputs, so gradients carry the same bits for any n (tests/test_trainer_gpu.py).
        self.checkpoint_layers = None
        # optional fn(layer_index, rows [B*L, h]) called with every decoder layer's output (what the reference's forced
        # `output_hidden_states=True`, metamorph_llama.py:345, would collect); rows of a left-padded batch are in the moved
        # (right-padded) layout.  Used by the full-depth parity test; None costs nothing.
        self.layer_output_hook = None

    def rope_tables(self, L, device):
        """cos / sin [>= L, head_dim] bf16 of the config's RoPE variant (metamorph_amd.rope: default, llama3, linear)."""
        if self._rope is None or self._rope[0][0] < L or self._rope[0][1] != str(device):
            Lc = max(L, 256)
            rp = self.rope
            cos, sin = ops.rope_table_freq(Lc, rp.head_dim, rp.inv_freq, rp.attention_scaling, device)
            self._rope = ((Lc, str(device)), cos, sin)
        return self._rope[1], self._rope[2]


class MetaMorphLlamaForCausalLM(PreTrainedModel, GenerationMixin, MetaMorphMetaForCausalLM):
    config_class = MetaMorphConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = True      # honoured as per-layer recompute (functional.LayerMeta.recompute)
    accepts_loss_kwargs = False                 # like the reference's forward(): no `num_items_in_batch` normalisation inside the model
    _no_split_modules = ["_DecoderLayer"]
    _keys_to_ignore_on_load_unexpected = [r"model\.vision_tower\.vision_tower\.head\..*", r".*rotary_emb\.inv_freq"]
    # config.tie_word_embeddings (LLaMA-3.2 1B / 3B bases): lm_head.weight IS model.embed_tokens.weight, as in HF's LlamaForCausalLM the
    # reference subclasses (metamorph_llama.py:223); post_init() / from_pretrained tie them through this mapping.  One Parameter, one
    # gradient buffer: the fused CE's weight gradient and the splice's embedding-row sums accumulate into it (functional.grad_target).
    _tied_weights_keys = {"lm_head.weight": "model.embed_tokens.weight"}

    def __init__(self, config, use_vision_ar=True, vision_head="None", vision_coef=1.0, normalize_vision=False,
                 apply_softmax=False, vision_delay_load=True, full_ar=False):
        super().__init__(config)
        self.model = MetaMorphLlamaModel(config, vision_delay_load=vision_delay_load)
        self.pretraining_tp = getattr(config, "pretraining_tp", 1)
        self.vocab_size = config.vocab_size
        self.lm_head = HipLinear(config.hidden_size, config.vocab_size, bias=False)
        self.normalize_vision = bool(normalize_vision) or bool(getattr(config, "normalize_vision", False))
        self.apply_softmax = bool(apply_softmax)
        vision_head = getattr(config, "vision_head_type", vision_head)
        # the reference hard-codes 1152 (metamorph_llama.py:255) = the SO400M feature width; here: the tower's feature width, i.e.
        # mm_hidden_size, or a quarter of it under 'concat_interpolation' (mm_hidden_size = 4 x tower width, siglip_encoder.py:110) --
        # so that reference checkpoints load unchanged in every configuration
        hv = getattr(config, "mm_hidden_size", 1152)
        if getattr(config, "image_token_reduction", "none") == "concat_interpolation":
            hv //= 4
        h = config.hidden_size
        if vision_head == "linear":
            self.vision_head = HipLinear(h, h)
        elif vision_head == "mlp":
            self.vision_head = nn.Sequential(HipLinear(h, h), HipGELU(), HipLinear(h, hv))
        elif vision_head == "mlp2x_gelu":
            self.vision_head = nn.Sequential(HipLinear(h, h), HipGELU(), HipLinear(h, h), HipGELU(), HipLinear(h, hv))
        else:
            self.vision_head = HipLinear(h, hv)
        self._vision_head_out = h if vision_head == "linear" else hv
        self.use_vision_ar = use_vision_ar
        self.vision_coef = vision_coef
        self._loss_language_t = None
        self._loss_image_ar_t = None
        self._mm_plan = None
        self._compact_hwm = {}                      # (B, L) -> compact decoder rows in use (constant across steps: llm_forward)
        self.post_init()

    # ------------------------------------------------------------------ HF plumbing
    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, HipLinear):
            nn.init.normal_(module.weight, mean=0.0, std=std)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, HipEmbedding):
            nn.init.normal_(module.weight, mean=0.0, std=std)

    def get_model(self):
        return self.model

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, value):
        self.lm_head = value

    def resize_token_embeddings(self, new_num_tokens=None, pad_to_multiple_of=None, mean_resizing=True):
        if new_num_tokens is None or new_num_tokens == self.config.vocab_size:
            return self.get_input_embeddings()
        tied = self.lm_head.weight is self.model.embed_tokens.weight
        for mod in (self.model.embed_tokens, self.lm_head):
            if tied and mod is self.lm_head:
                mod.weight = self.model.embed_tokens.weight          # stays ONE parameter
                break
            old = mod.weight.data
            new = torch.empty((new_num_tokens, old.shape[1]), device=old.device, dtype=old.dtype)
            nn.init.normal_(new, std=getattr(self.config, "initializer_range", 0.02))
            n = min(old.shape[0], new_num_tokens)
            new[:n] = old[:n]
            mod.weight = nn.Parameter(new, requires_grad=mod.weight.requires_grad)
        self.model.embed_tokens.num_embeddings = new_num_tokens
        self.lm_head.out_features = new_num_tokens
        self.config.vocab_size = self.vocab_size = new_num_tokens
        return self.get_input_embeddings()

    @property
    def loss_language(self):
        """float, like the reference's attribute (metamorph_llama.py:464); the device->host read happens here,
        when a logger asks, not inside forward."""
        return float("nan") if self._loss_language_t is None else float(self._loss_language_t)

    @property
    def loss_image_ar(self):
        return float("nan") if self._loss_image_ar_t is None else float(self._loss_image_ar_t)

    # ------------------------------------------------------------------ plan handling
    def _plan_for(self, inputs_embeds, attention_mask, labels, image_positions):
        cached = self._mm_plan
        if cached is not None and cached[0] is inputs_embeds:
            self._mm_plan = None
            return cached[1]
        # inputs_embeds supplied by the caller: derive the index arrays from the given tensors (one host copy)
        B, L, _ = inputs_embeds.shape
        msk = np.ones((B, L), dtype=bool) if attention_mask is None else host_array(attention_mask).astype(bool)
        left = getattr(self.config, "tokenizer_padding_side", "right") == "left"
        seqlens = msk.sum(1).astype(np.int32)
        if not all((msk[b, L - seqlens[b]:] if left else msk[b, : seqlens[b]]).all() for b in range(B)):
            raise NotImplementedError("attention_mask must be a contiguous padding mask on the configured tokenizer_padding_side")
        lab = host_array(labels)
        pos = np.zeros((B, L), dtype=np.int64) if image_positions is None else host_array(image_positions)
        nxt = np.zeros((B, L), dtype=bool)
        nxt[:, :-1] = pos[:, 1:] == 1
        st = ce_rows = None
        n_valid = 0
        if lab is not None:
            s = np.full((B, L), IGNORE_INDEX, dtype=np.int64)
            s[:, :-1] = lab[:, 1:]
            st = s.reshape(-1).astype(np.int32)
            ce_rows = np.flatnonzero(st != IGNORE_INDEX).astype(np.int32)
            n_valid = int(ce_rows.shape[0])
        z = np.zeros(0, dtype=np.int32)
        plan = SplicePlan(B=B, L=L, rows_per_image=0, src=z, labels=lab, attention_mask=msk, image_positions=pos,
                          position_ids=np.zeros((B, L), dtype=np.int64), seqlens=seqlens, target_keep=np.zeros(0, dtype=np.int64),
                          feat_row=z, pred_rows=np.flatnonzero(nxt.reshape(-1)).astype(np.int32), shift_targets=st,
                          ce_rows=ce_rows, n_valid=n_valid, emb_tok=z, emb_seg=np.zeros(1, dtype=np.int32), emb_pos=z,
                          images_consumed=0, padding_side="left" if left else "right")
        return upload_plan(plan, inputs_embeds.device)

    # ------------------------------------------------------------------ the hot path
    def llm_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                    inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                    return_dict=None, cache_position=None, image_positions=None, decoding=False, image_features=None):
        """Reference metamorph_llama.py:285-498."""
        if past_key_values is not None or use_cache:
            # HF `generate()` (reference :711-717 -> GenerationMixin -> forward with a cache): prompt pass / one row per step on the
            # decode-shape kernels
            if labels is not None or decoding:
                raise NotImplementedError("past_key_values / use_cache together with labels or the image-AR head: the cached path is the "
                                          "generation path (no loss)")
            # images handed through prepare_inputs_for_generation arrive spliced, with image_positions / image_features: without labels the
            # reference computes no loss from them either (metamorph_llama.py:420-474 sits under `if labels is not None`), so they are dropped
            return self._cached_forward(input_ids, inputs_embeds, attention_mask, past_key_values, return_dict)
        if output_attentions:
            raise NotImplementedError("output_attentions: attention probabilities are never materialised by the flash kernel")
        return_dict = True if return_dict is None else return_dict
        cfg = self.config
        F.params_ready(None)
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        if inputs_embeds.dtype != BF16:
            raise TypeError(f"inputs_embeds must be bf16, got {inputs_embeds.dtype}")
        B, L, h = inputs_embeds.shape
        pd = self._plan_for(inputs_embeds, attention_mask, labels, image_positions)
        plan = pd["host"]
        dev = inputs_embeds.device
        if position_ids is not None:
            # The kernels rotate row l of a sample by l (+ the sample's left-padding offset); RoPE only sees position DIFFERENCES, so any
            # position_ids that advance by one over a sample's valid rows (what the reference builds, metamorph_arch.py:362-399, and
            # what HF's default arange is) give the same attention.  Anything else (packed sequences, gaps) is refused, not ignored.
            pid = position_ids.detach().cpu().numpy().reshape(B, L)
            for b in range(B):
                rows = np.flatnonzero(plan.attention_mask[b])
                if rows.size > 1 and not (np.diff(pid[b, rows]) == 1).all():
                    raise NotImplementedError("position_ids must advance by one over each sample's valid rows (custom positions are not supported)")
        Hq, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
        d = _head_dim(cfg)
        # Left padding (tokenizer_padding_side = "left", reference metamorph_arch.py:362-386): the attention kernels take per-sample
        # lengths counted from row 0, so the batch is moved to the right-padded row layout for the decoder and back afterwards; the
        # RoPE position of a moved row stays its ORIGINAL row index (HF: position_ids = arange(L), padding included), which the
        # kernels get as a per-sample offset.  Causal attention over the valid rows is unchanged by the move.
        shift = plan.padding_side == "left" and not plan.attention_mask.all()
        pos_off = None
        if shift:
            to_right, to_left, off = _left_pad_maps(plan.seqlens, B, L)
            moved = torch.from_numpy(np.concatenate([to_right, to_left, off])).to(dev)
            to_right_d, to_left_d, pos_off = moved[:B * L], moved[B * L:2 * B * L], moved[2 * B * L:]
        cos, sin = self.model.rope_tables(2 * L if shift else L, dev)
        # Padding-free rows (reference: batches are right-padded to their longest sample, train.py:1258-1284, and the intended attention is
        # varlen, llama_flash_attn_monkey_patch.py:95-104; tokens/s counts VALID tokens): when enough of the B x L rows are
        # padding, the decoder runs on the sum(len) valid rows only (LayerMeta.c2p / p2c) -- norms, GEMMs and SwiGLU never touch a padding
        # row; q|k|v -> attention -> o visits the padded layout through row gathers.  `mm355_compact_rows`: True / False / "auto" (default:
        # on when at least an eighth of the rows is saved; the gathers + one RoPE pass per layer cost ~2 % of a layer).
        c2p_d = p2c_d = None
        mode = getattr(cfg, "mm355_compact_rows", "auto")
        n_valid_rows = int(plan.seqlens.sum())
        if mode is not False and n_valid_rows > 0:
            # The compact row count is kept CONSTANT across steps of one (B, L) shape: a high-water mark of what the batches seen so far needed,
            # in steps of 1/16 of B x L.  Tensor sizes that change from step to step defeat the caching allocator at the occupancy a
            # training run has (222 of 288 GB at B = 16): every new maximum costs a flush of the cache, and PyTorch-ROCm has no expandable
            # segments -- measured (profiles/r6_ragged_compact_rows.log): per-batch row counts at 5 - 20 % padding ran 1.4 x SLOWER than the
            # padded layout.  With a constant count the shapes are as static as the padded path's and the saving is that of the fullest batch.
            step_rows = max(256, (B * L // 16 + 255) // 256 * 256)
            need = (n_valid_rows + step_rows - 1) // step_rows * step_rows
            hwm = self._compact_hwm.get((B, L), 0)
            rows = min(max(need, hwm), (B * L + 255) // 256 * 256)
            if mode == "exact":                                      # this batch's own count (256-row granule): sizes change from step to step
                rows = min((n_valid_rows + 255) // 256 * 256, (B * L + 255) // 256 * 256)
            if mode is True or mode == "exact" or rows <= 0.875 * B * L:   # "auto": at least an eighth of the rows saved
                if mode != "exact":
                    self._compact_hwm[(B, L)] = rows
                c2p_d, p2c_d = pd["c2p"][:rows], pd["p2c"]            # uploaded with the rest of the plan (one pinned, asynchronous copy)
        self._decoder_rows = (int(c2p_d.shape[0]) if c2p_d is not None else B * L, B * L)     # (rows the decoder ran on, padded rows): introspection
        meta = F.LayerMeta(B, L, Hq, Hkv, d, cfg.intermediate_size, cfg.rms_norm_eps, cos, sin, pd["seqlens"],
                           recompute=bool(self.model.gradient_checkpointing) and self.training, pos_offset=pos_off, c2p=c2p_d, p2c=p2c_d)

        x = inputs_embeds.reshape(B * L, h)
        if not x.is_contiguous():
            x = x.contiguous()
        if shift:
            x = F.RowsPermuteFn.apply(x, to_right_d, to_left_d)
        if c2p_d is not None:
            x = F.RowsPermuteFn.apply(x, c2p_d, p2c_d)                  # [rows_compact, h]; backward gathers with p2c (padding rows: zero)
        tap = self.model.layer_output_hook
        n_ck = self.model.checkpoint_layers
        meta_keep = meta
        if meta.recompute and n_ck is not None:
            meta_keep = copy.copy(meta)
            meta_keep.recompute = False
        for li, layer in enumerate(self.model.layers):
            x = F.decoder_layer(x, layer, meta if (n_ck is None or li < int(n_ck)) else meta_keep)
            if tap is not None:                                         # the reference's output_hidden_states tuple, one entry at a time
                tap(li, x if p2c_d is None else ops.rows_gather(x.detach(), p2c_d))
        if c2p_d is not None:
            x = F.RowsPermuteFn.apply(x, p2c_d, c2p_d)                  # back to [B * L, h] (padding rows: zeros, as the padded path leaves them
        if shift:
            x = F.RowsPermuteFn.apply(x, to_left_d, to_right_d)
        hid = self.model.norm(x)                                       # [B*L, h]
        hidden_states = hid.view(B, L, h)

        pred_z = None
        if decoding:                                                    # metamorph_llama.py:363-377
            with torch.no_grad():
                last = hidden_states[:, -1, :].contiguous()
                pred_z = self.vision_head(last)
                if self.normalize_vision:
                    pred_z = ops.bilinear_l2norm(pred_z.view(B, 1, -1).contiguous(), 1, 1, True).view(B, -1)
                if self.apply_softmax:
                    pred_z = ops.softmax_rows(pred_z.contiguous(), 0.07)
                prediction = self.model.mm_projector(pred_z)
                hidden_states = hidden_states.clone()
                hidden_states[:, -1, :] = prediction
                hid = hidden_states.view(B * L, h)

        loss = None
        logits = None
        tp = int(getattr(cfg, "pretraining_tp", 1) or 1)
        if tp > 1 and self.lm_head.weight.shape[0] % tp:
            # reference :393-396: `weight.split(vocab_size // tp)` yields tp + 1 slices when tp does not divide the vocabulary and the
            # loop over range(tp) silently drops the tail rows (for V = 128258 and tp = 4: <image_start> / <image_end>); not reproduced
            raise NotImplementedError(f"pretraining_tp={tp} does not divide the vocabulary ({self.lm_head.weight.shape[0]} rows)")
        # pretraining_tp > 1 (reference :393-396) computes the logits in `tp` vocabulary slices and concatenates them: every logit is
        # the same K-long dot product either way, and the kernels below tile the vocabulary anyway (the fused CE walks row chunks x
        # 256-column tiles), so the sliced and the unsliced form are one computation here (golden: tests/golden/r3_pretraining_tp2_*)
        if labels is None or getattr(cfg, "mm355_return_logits", False) or not torch.is_grad_enabled():
            # inference / evaluation: the full fp32 logits tensor of the reference (metamorph_llama.py:398-399)
            logits = ops.gemm(hid.detach(), self.lm_head.weight.data, out_f32=True).view(B, L, -1)
        if labels is not None:
            nan = torch.full((), float("nan"), device=dev, dtype=torch.float32)
            if plan.n_valid > 0:
                ce = F.LinearCrossEntropyFn.apply(hid, self.lm_head.weight, self.lm_head, pd, plan.n_valid)
            else:
                ce = nan                                                # mean over no targets (torch CE semantics)
            loss = ce
            if image_positions is not None:
                if image_features is not None:
                    R = int(plan.pred_rows.shape[0])
                    tgt = image_features.reshape(-1, image_features.shape[-1])
                    cosine = self.normalize_vision and not self.apply_softmax
                    Rt = tgt.shape[0]
                    if R == 0 and Rt == 0 and (cosine or self.apply_softmax):
                        l_img = nan          # mean over an empty tensor -- SURVEY A9 (understanding-only / text-only batches)
                    elif cosine and (R != Rt or tgt.shape[1] != self._vision_head_out):
                        # F.cosine_similarity raises on a row-count mismatch -- or, under 'concat_interpolation', on the width mismatch
                        # between the 1152-wide head and the 4 x 1152 targets -- and the reference's try/except (:451-455) substitutes CE
                        l_img = ce
                    elif self.apply_softmax and (R != Rt or R == 0):
                        # soft-CE: `target * log(pred)` with different row counts is a broadcasting error in the reference (:445)
                        raise RuntimeError(f"image-AR head (soft-CE): {R} prediction rows vs {Rt} target rows -- the reference raises here "
                                           f"as well (metamorph_llama.py:445)")
                    elif R == 0 or Rt == 0:
                        # mean-abs (`mse_loss_fn`, :211-219): no target rows -> division by len(z) = 0; no prediction rows -> the Python
                        # float 0.0 has no .item() (:466)
                        raise RuntimeError(f"image-AR head (mean-abs): {R} prediction rows vs {Rt} target rows -- the reference raises here "
                                           f"as well (metamorph_llama.py:211-219,466)")
                    else:
                        pred_in = F.RowsGatherFn.apply(hid, pd["pred_rows"])
                        pred = self.vision_head(pred_in).contiguous()
                        tgt = tgt.to(BF16).contiguous()
                        if self.apply_softmax:                           # soft-CE against softmax(feat / 0.07) targets (:437-447)
                            l_img = F.SoftCELossFn.apply(pred, tgt, self.normalize_vision)
                        elif self.normalize_vision:                      # -mean cos (:449-455)
                            l_img = F.CosineLossFn.apply(pred, tgt, True)
                        else:
                            # mean |t - p| (`mse_loss_fn`, :459) -- the constructor default.  The reference walks zip(target rows,
                            # prediction rows) and divides by len(target): with R != Rt it silently uses the first min(R, Rt) row pairs
                            # and still divides by Rt (:215-217)
                            n = min(R, Rt)
                            if n != R:
                                pred = pred[:n].contiguous()
                            l_img = F.MeanAbsLossFn.apply(pred, tgt[:n].contiguous() if n != Rt else tgt, Rt)
                else:
                    l_img = ce                                          # metamorph_llama.py:461-462
                self._loss_language_t = ce.detach()
                self._loss_image_ar_t = l_img.detach()
                if self.use_vision_ar:
                    # `if loss_image_ar.item() != 0` without the host sync (NaN != 0 is True, so NaN propagates)
                    loss = torch.where(l_img.detach() != 0, ce + self.vision_coef * l_img, ce)

        if not return_dict:
            out = (logits,)
            return (loss,) + out if loss is not None else out
        return CausalLMOutputWithPast(loss=pred_z if decoding else loss, logits=logits, past_key_values=None,
                                      hidden_states=hidden_states, attentions=None)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                image_sizes=None, return_dict=None, cache_position=None, image_embeds=None):
        """Reference metamorph_llama.py:603-660 (same parameter list: no **kwargs, so HF Trainer keeps dividing the loss by the
        gradient-accumulation steps exactly as it does for the reference class)."""
        image_positions = None
        target = None
        if inputs_embeds is None and past_key_values is not None and (images is None or past_key_values.get_seq_length() > 0):
            # a decode step under HF generate(): token ids only, the prompt (with its images) already sits in the cache -- the
            # reference reaches the same early return in prepare_inputs_labels_for_multimodal (`input_ids.shape[1] == 1`, :187-193)
            pass
        elif inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels, image_positions,
             target) = self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                                 labels, images, image_sizes, image_embeds)
        return self.llm_forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                                use_cache=use_cache, output_attentions=output_attentions,
                                output_hidden_states=output_hidden_states, return_dict=return_dict,
                                image_positions=image_positions, image_features=target)

    # ------------------------------------------------------------------ decoding (reference :502-597, :665-717)
    @torch.no_grad()
    def greedy_decode(self, position_ids, attention_mask, inputs_embeds, start_image_token_id=DEFAULT_IMAGE_START_ID,
                      end_image_token_id=DEFAULT_IMAGE_END_ID, eos_token_id=(128001, 128009), do_sample=None,
                      temperature=None, top_p=None, num_beams=None, max_new_tokens=1024, use_cache=None, output_image=False):
        """Token mode / continuous 'image mode' greedy loop.  use_cache=False re-runs the prefix every step exactly like
        the reference (which forces use_cache=False, O(L^2)); the default keeps a KV cache and feeds one row per step
        through the decode-shape kernels -- same state machine, same outputs."""
        if use_cache is None or use_cache:
            return self._greedy_decode_cached(inputs_embeds, start_image_token_id, end_image_token_id, eos_token_id,
                                              max_new_tokens, output_image)
        in_image_mode = False
        generated, image_embeds = [], []
        total_image_tokens = 0
        total_out = 0
        num_image_tokens = self.get_model().vision_tower.image_token_len
        eos = set(eos_token_id)
        while True:
            out = self.llm_forward(inputs_embeds=inputs_embeds, attention_mask=None, return_dict=True, decoding=in_image_mode)
            image_embed = out.loss
            next_token = int(torch.argmax(out.logits[:, -1, :], dim=-1)[0])
            next_embed = out.hidden_states[:, -1, :].unsqueeze(0)
            tok_embed = self.model.embed_tokens(torch.tensor([[next_token]], device=inputs_embeds.device))
            if (not in_image_mode) and next_token == start_image_token_id:
                in_image_mode = True
                generated.append(next_token)
                inputs_embeds = torch.cat((inputs_embeds, tok_embed), dim=1)
            elif in_image_mode and total_image_tokens < num_image_tokens:
                total_image_tokens += 1
                image_embeds.append(image_embed)
                inputs_embeds = torch.cat((inputs_embeds, next_embed), dim=1)
                if total_image_tokens == num_image_tokens:
                    in_image_mode = False
            elif next_token == end_image_token_id:
                in_image_mode = False
                total_image_tokens = 0
                generated.append(next_token)
                inputs_embeds = torch.cat((inputs_embeds, tok_embed), dim=1)
            else:
                inputs_embeds = torch.cat((inputs_embeds, tok_embed), dim=1)
                generated.append(next_token)
            total_out += 1
            if next_token in eos or total_out > max_new_tokens:
                break
        emb = torch.cat(image_embeds, dim=0) if image_embeds else torch.tensor([], dtype=torch.float32, device=inputs_embeds.device)
        output = [torch.tensor(generated, dtype=torch.int32, device=inputs_embeds.device)]
        return (output, emb) if output_image else output

    # ------------------------------------------------------------------ cached decode (row N1)
    def _decode_meta(self, L):
        cfg = self.config
        h, Hq, Hkv = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads
        return h, F.LayerMeta(1, L, Hq, Hkv, _head_dim(cfg), cfg.intermediate_size, cfg.rms_norm_eps, None, None, None)

    @torch.no_grad()
    def _head_row(self, x, in_image_mode):
        """Final norm + (image mode: vision_head -> normalize -> mm_projector) + fp32 logits of ONE hidden row
        (reference metamorph_llama.py:363-377, 398-399).  Returns (logits [1,V] f32, row fed back in image mode, pred_z)."""
        hid = self.model.norm(x)
        pred_z = None
        if in_image_mode:
            pred_z = self.vision_head(hid)
            if self.normalize_vision:
                pred_z = ops.bilinear_l2norm(pred_z.view(1, 1, -1).contiguous(), 1, 1, True).view(1, -1)
            if self.apply_softmax:
                pred_z = ops.softmax_rows(pred_z.contiguous(), 0.07)
            hid = self.model.mm_projector(pred_z)
        logits = ops.gemv(hid.contiguous(), self.lm_head.weight.data, out=torch.empty((1, self.lm_head.weight.shape[0]), device=x.device,
                                                                                    dtype=torch.float32))
        return logits, hid, pred_z

    @torch.no_grad()
    def _greedy_decode_cached(self, inputs_embeds, start_image_token_id, end_image_token_id, eos_token_id, max_new_tokens,
                              output_image):
        if inputs_embeds.shape[0] != 1:
            raise NotImplementedError("greedy_decode handles one sequence (as the reference's loop does)")
        if inputs_embeds.dtype != BF16:
            raise TypeError(f"inputs_embeds must be bf16, got {inputs_embeds.dtype}")
        dev = inputs_embeds.device
        F.params_ready(None)
        L0 = inputs_embeds.shape[1]
        num_image_tokens = self.get_model().vision_tower.image_token_len
        h, meta = self._decode_meta(L0)
        max_len = L0 + max_new_tokens + 2
        cos, sin = self.model.rope_tables(max_len, dev)
        meta.cos, meta.sin = cos, sin
        cache = F.KVCache(len(self.model.layers), max_len, meta.Hkv * meta.d, dev, Hq=meta.Hq, d=meta.d)
        x = F.decoder_prefill(inputs_embeds.reshape(L0, h).contiguous(), self.model.layers, meta, cache)[-1:].contiguous()
        stepper = F.DecodeStepGraph(self.model.layers, meta, cache, cos, sin, h, dev)
        in_image_mode = False
        generated, image_embeds = [], []
        total_image_tokens = 0
        total_out = 0
        eos = set(eos_token_id)
        while True:
            logits, fed_back, pred_z = self._head_row(x, in_image_mode)
            next_token = int(torch.argmax(logits[0], dim=-1))
            if (not in_image_mode) and next_token == start_image_token_id:
                in_image_mode = True
                generated.append(next_token)
                row = None
            elif in_image_mode and total_image_tokens < num_image_tokens:
                total_image_tokens += 1
                image_embeds.append(pred_z)
                row = fed_back
                if total_image_tokens == num_image_tokens:
                    in_image_mode = False
            elif next_token == end_image_token_id:
                in_image_mode = False
                total_image_tokens = 0
                generated.append(next_token)
                row = None
            else:
                generated.append(next_token)
                row = None
            total_out += 1
            if next_token in eos or total_out > max_new_tokens:
                break
            if row is None:
                row = self.model.embed_tokens(torch.tensor([[next_token]], device=dev)).view(1, h)
            x = stepper.step(row)
        emb = torch.cat(image_embeds, dim=0) if image_embeds else torch.tensor([], dtype=torch.float32, device=dev)
        output = [torch.tensor(generated, dtype=torch.int32, device=dev)]
        return (output, emb) if output_image else output

    # ------------------------------------------------------------------ HF generate() on the decode kernels (reference :711-738)
    def _prefill_batch(self, seqs, cache):
        """Prompt pass of a batch: seqs = one [L_b, h] tensor of prompt rows per sequence (padding already stripped) -> their hidden rows
        (pre final norm), one [L_b, h] tensor each; allocates and fills `cache.kv` and captures the per-token step.  Prompts of one length
        (the beams of a beam search, an unpadded batch) go through the decoder as ONE batch; ragged prompts one after the other."""
        dev = seqs[0].device
        B, h = len(seqs), seqs[0].shape[1]
        lens = [int(x.shape[0]) for x in seqs]
        L0 = max(lens)
        cap = cache.capacity if cache.capacity is not None else L0 + 1024 + 2
        if cap < L0 + 1:
            raise ValueError(f"HipKVCache capacity {cap} is smaller than the prompt ({L0} rows)")
        cos, sin = self.model.rope_tables(cap, dev)
        _, meta = self._decode_meta(L0)
        meta.cos, meta.sin = cos, sin
        cache.kv = F.KVCache(len(self.model.layers), cap, meta.Hkv * meta.d, dev, Hq=meta.Hq, d=meta.d, batch=B)
        cache.meta = meta
        if B > 1 and all(n == L0 for n in lens):
            _, mb = self._decode_meta(L0)
            mb.B, mb.cos, mb.sin = B, cos, sin
            rows = F.decoder_prefill(torch.cat(seqs, 0), self.model.layers, mb, cache.kv)
            out = list(rows.view(B, L0, h).unbind(0))
        else:
            out = []
            for b, x2d in enumerate(seqs):
                _, mb = self._decode_meta(lens[b])
                mb.cos, mb.sin = cos, sin
                out.append(F.decoder_prefill(x2d.contiguous(), self.model.layers, mb, cache.kv, row=b))
        cache.stepper = F.DecodeStepGraph(self.model.layers, meta, cache.kv, cos, sin, h, dev)
        return out

    def _decode_batch(self, x, cache):
        """New rows x [B, n, h] (usually n = 1) appended against the batch's cache, every position in ONE pass for all B sequences ->
        their hidden rows [B, n, h] (pre final norm)."""
        outs = [cache.stepper.step(x[:, t].contiguous()).clone() for t in range(x.shape[1])]
        return outs[0].unsqueeze(1) if len(outs) == 1 else torch.stack(outs, 1)

    def _rows_logits(self, rows, return_hidden=False):
        """final norm + lm_head -> fp32 logits [n, V] (reference :349-359 final norm, :393-399); return_hidden: (logits, normed rows)."""
        hid = self.model.norm(rows)
        if hid.shape[0] <= 32 and ops.gemv_supported(hid, self.lm_head.weight.data):     # (17 .. 32 rows: wide weights only -- the lm_head is one)
            logits = ops.gemv(hid.contiguous(), self.lm_head.weight.data,
                              out=torch.empty((hid.shape[0], self.lm_head.weight.shape[0]), device=rows.device, dtype=torch.float32))
        else:
            logits = ops.gemm(hid, self.lm_head.weight.data, out_f32=True)
        return (logits, hid) if return_hidden else logits

    @torch.no_grad()
    def _cached_forward(self, input_ids, inputs_embeds, attention_mask, past_key_values, return_dict):
        F.params_ready(None)
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        if inputs_embeds.dtype != BF16:
            raise TypeError(f"inputs_embeds must be bf16, got {inputs_embeds.dtype}")
        B, n, h = inputs_embeds.shape
        cache = past_key_values
        if cache is None:
            cache = HipKVCache()
        if not isinstance(cache, HipKVCache):
            if cache.get_seq_length() != 0:
                raise NotImplementedError("past_key_values must be a metamorph_amd HipKVCache (generate() creates one); a foreign, "
                                          "already filled transformers Cache holds tensors in another layout")
            cache = HipKVCache()                              # an empty HF cache object: swap in ours
        first = cache.kv is None
        if first:
            # batch rows / beams are independent sequences of ONE cache.  A batch of prompts of different lengths arrives LEFT-padded with
            # its attention mask (the reference: HF generate derives position_ids = cumsum(mask) - 1 from it, so every row is decoded
            # exactly as it would be alone).  Here the padding rows are never computed or cached: each sequence keeps its own length,
            # `pads` only restores the common length HF counts.
            pads = [0] * B
            if attention_mask is not None and not bool(attention_mask.to(torch.bool).all()):
                m = attention_mask.to(torch.bool).cpu()
                if tuple(m.shape) != (B, n):
                    raise ValueError(f"attention_mask {tuple(m.shape)} does not match the prompt batch {(B, n)}")
                for b in range(B):
                    nb = int(m[b].sum())
                    if nb == 0 or not bool(m[b, n - nb:].all()):
                        raise NotImplementedError("cached decoding takes LEFT-padded prompts (valid rows at the end); right padding puts pad rows "
                                                  "between the prompt and the generated tokens in the reference as well")
                    pads[b] = n - nb
            cache.pads = pads
            outs = self._prefill_batch([inputs_embeds[b, pads[b]:].reshape(n - pads[b], h) for b in range(B)], cache)
            # padding positions: rows nobody reads (HF takes logits[:, -1])
            rows = torch.cat([r if not pads[b] else torch.cat([r.new_zeros((pads[b], h)), r], 0) for b, r in enumerate(outs)], 0)
        else:
            if len(cache.pads) != B:
                raise ValueError(f"cache holds {len(cache.pads)} sequences, the step brings {B}")
            rows = self._decode_batch(inputs_embeds, cache).reshape(B * n, h)
        logits, hidden = self._rows_logits(rows.contiguous(), return_hidden=True)
        logits, hidden = logits.view(B, n, -1), hidden.view(B, n, h)
        if return_dict is False:
            return (logits, cache)
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=cache, hidden_states=hidden, attentions=None)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        """Reference metamorph_llama.py:721-735: the HF default plus `images` / `image_sizes` handed through."""
        images = kwargs.pop("images", None)
        image_sizes = kwargs.pop("image_sizes", None)
        inputs = super().prepare_inputs_for_generation(input_ids, past_key_values=past_key_values, inputs_embeds=inputs_embeds, **kwargs)
        if images is not None:
            inputs["images"] = images
        if image_sizes is not None:
            inputs["image_sizes"] = image_sizes
        return inputs

    @torch.no_grad()
    def generate(self, inputs=None, images=None, image_sizes=None, output_image=False, use_customize_greedy=True,
                 image_embeds=None, **kwargs):
        position_ids = kwargs.pop("position_ids", None)
        attention_mask = kwargs.pop("attention_mask", None)
        if images is not None or image_embeds is not None:
            (_, position_ids, attention_mask, _, inputs_embeds, _, _, _) = self.prepare_inputs_labels_for_multimodal(
                inputs, position_ids, attention_mask, None, None, images, image_sizes=image_sizes, image_embeds=image_embeds)
        else:
            inputs_embeds = self.get_model().embed_tokens(inputs)
        if not use_customize_greedy:
            # reference :711-717: `super().generate(position_ids=..., attention_mask=..., inputs_embeds=..., **kwargs)` -- HF's sampling
            # / greedy search driving forward() with a cache.  Here the cache is a HipKVCache (decode kernels + hipGraph replay).
            if kwargs.get("past_key_values") is None and kwargs.get("use_cache", True):
                L0 = int(inputs_embeds.shape[1])
                new = kwargs.get("max_new_tokens")
                if new is None and kwargs.get("max_length") is not None:
                    new = max(int(kwargs["max_length"]) - L0, 1)         # HF subtracts the inputs_embeds length from max_length itself
                if new is None:
                    gc = kwargs.get("generation_config") or self.generation_config
                    new = getattr(gc, "max_new_tokens", None)
                    if new is None:                                      # HF's max_length is a TOTAL that includes the inputs_embeds rows
                        new = max(int(getattr(gc, "max_length", None) or 20) - L0, 1)
                kwargs["past_key_values"] = HipKVCache(capacity=L0 + int(new) + 2)
            return GenerationMixin.generate(self, position_ids=position_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                                            **kwargs)
        return self.greedy_decode(position_ids=position_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                                  output_image=output_image, **kwargs)


for _reg, _args in ((AutoConfig.register, ("metamorph_llama", MetaMorphConfig)),
                    (AutoModelForCausalLM.register, (MetaMorphConfig, MetaMorphLlamaForCausalLM))):
    try:
        _reg(*_args)
    except ValueError:      # already registered (e.g. the reference package imported in the same process)
        pass
