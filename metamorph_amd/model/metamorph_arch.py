"""Mixins with the reference's names (reference metamorph/model/metamorph_arch.py:21-469).

`prepare_inputs_labels_for_multimodal` keeps the reference's signature and 8-tuple return, but the
per-sample Python loop of device ops (one sync per sample, O(B*segments) tiny kernels) is replaced by:
one host copy of the [B,T] id/label/mask arrays -> `build_splice_plan` (bit-exact integer bookkeeping)
-> one gather kernel forward and deterministic scatter kernels backward.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import functional as F
from .. import ops
from ..constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IMAGE_START_ID
from ..hostmirror import host_array
from ..splice_plan import SplicePlan, build_splice_plan, compact_row_maps
from .modules import HipLinear
from .multimodal_encoder.builder import build_vision_tower
from .multimodal_projector.builder import build_vision_projector

BF16 = torch.bfloat16


class MetaMorphMetaModel:
    """Builds the tower and the projector next to the LLaMA stack (reference metamorph_arch.py:21-96)."""

    def _init_vision(self, config, vision_delay_load=True):
        if hasattr(config, "mm_vision_tower"):
            self.vision_tower = build_vision_tower(config, delay_load=vision_delay_load)
            self.mm_projector = build_vision_projector(config)
            # the reference carries an unused Linear(4096, hidden) (metamorph_arch.py:31); kept so checkpoints round-trip
            self.vision_proj = HipLinear(4096, config.hidden_size)
            self.temperature_in = 0.1

    def get_vision_tower(self):
        vt = getattr(self, "vision_tower", None)
        return vt[0] if type(vt) is list else vt

    def initialize_vision_modules(self, model_args, fsdp=None):
        vision_tower = model_args.vision_tower
        self.config.mm_vision_tower = vision_tower
        if self.get_vision_tower() is None:
            vt = build_vision_tower(model_args)
            self.vision_tower = [vt] if fsdp is not None and len(fsdp) > 0 else vt
        else:
            vt = self.vision_tower[0] if fsdp is not None and len(fsdp) > 0 else self.vision_tower
            if not vt.is_loaded:
                vt.load_model()
        self.config.use_mm_proj = True
        self.config.mm_projector_type = getattr(model_args, "mm_projector_type", "linear")
        self.config.mm_hidden_size = vt.hidden_size
        self.config.mm_vision_select_layer = model_args.mm_vision_select_layer
        self.config.mm_vision_select_feature = model_args.mm_vision_select_feature
        self.config.mm_patch_merge_type = getattr(model_args, "mm_patch_merge_type", "flat")
        if getattr(self, "mm_projector", None) is None:
            self.mm_projector = build_vision_projector(self.config)
        else:
            for p in self.mm_projector.parameters():
                p.requires_grad = True
        self.vision_proj = HipLinear(4096, self.config.hidden_size)
        self.temperature_in = 1
        adapter = getattr(model_args, "pretrain_mm_mlp_adapter", None)
        if adapter is not None:
            weights = torch.load(adapter, map_location="cpu")
            self.mm_projector.load_state_dict({k.split("mm_projector.")[1]: v for k, v in weights.items() if "mm_projector" in k})


class PlanOnDevice(dict):
    """int32 index arrays of a SplicePlan uploaded once (one pinned staging copy) + the host plan itself."""


class _PinnedStage:
    """Two rotating pinned staging buffers (grow-only) for the per-step plan upload: no host allocation per step, and a buffer is reused
    only after the copy that last read it has finished (an event per buffer; by then two steps have passed)."""

    def __init__(self):
        self.bufs = [None, None]
        self.events = [None, None]
        self.i = 0

    def take(self, nbytes):
        self.i ^= 1
        ev = self.events[self.i]
        if ev is not None:
            ev.synchronize()
        b = self.bufs[self.i]
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
            self.bufs[self.i] = b
        return b

    def sent(self):
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.i] = ev


_STAGE = _PinnedStage()


def upload_plan(plan: SplicePlan, device, extra=None) -> PlanOnDevice:
    """Every array the step needs from the host plan in ONE pinned, asynchronous host -> device copy: the int32 index arrays, and (extra)
    the [B, L] tensors handed back to the caller (labels, attention mask, image positions, position ids) in their own dtypes."""
    names = ["src", "feat_row", "pred_rows", "seqlens", "emb_tok", "emb_seg", "emb_pos"]
    arrays = {n: getattr(plan, n).astype(np.int32, copy=False) for n in names}
    M = plan.B * plan.L
    if plan.ce_rows is not None:
        arrays["ce_rows"] = plan.ce_rows
        arrays["ce_targets"] = plan.shift_targets[plan.ce_rows].astype(np.int32)
        inv = np.full(M, -1, dtype=np.int32)
        inv[plan.ce_rows] = np.arange(plan.ce_rows.shape[0], dtype=np.int32)
        arrays["ce_inv"] = inv
    # padding-free decoder rows (ragged batches): compact <-> padded row maps over the right-padded layout the decoder runs in
    arrays["c2p"], arrays["p2c"] = compact_row_maps(plan.seqlens, plan.B, plan.L, full=True)
    for k, v in (extra or {}).items():
        if v is not None:
            arrays["x_" + k] = np.ascontiguousarray(v)
    offs, total = {}, 0
    for k, v in arrays.items():
        offs[k] = total
        total += (v.nbytes + 15) // 16 * 16                  # keep every array 16-byte aligned
    cuda = torch.device(device).type == "cuda"
    if cuda:
        stage = _STAGE.take(total)
        host = stage.numpy()
    else:
        host = np.zeros(max(total, 16), dtype=np.uint8)
    for k, v in arrays.items():
        host[offs[k]: offs[k] + v.nbytes] = v.reshape(-1).view(np.uint8)
    if cuda:
        t = stage[:max(total, 16)].to(device, non_blocking=True)
        _STAGE.sent()
    else:
        t = torch.from_numpy(host)
    out = PlanOnDevice()
    for k, v in arrays.items():
        td = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.bool_): torch.bool,
              np.dtype(np.uint8): torch.uint8}[v.dtype]
        out[k] = t[offs[k]: offs[k] + v.nbytes].view(td).view(v.shape)
    out["host"] = plan
    return out


class MetaMorphMetaForCausalLM(ABC):

    @abstractmethod
    def get_model(self):
        pass

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    # ------------------------------------------------------------------ A4
    def encode_images(self, images, return_prob=False):
        """tower -> mm_projector; second return = detached regression targets (reference metamorph_arch.py:140-164)."""
        image_features = self.get_model().get_vision_tower()(images)
        if return_prob:                                      # 'mlpsoftmax' connector taken apart (metamorph_arch.py:143-158)
            F.params_ready(None)
            linear1, softmax, linear2 = self.get_model().mm_projector[0], self.get_model().mm_projector[1], self.get_model().mm_projector[2]
            n, t, c = image_features.shape
            z = linear1(image_features.reshape(n * t, c).to(BF16).contiguous())
            target_prob = softmax(z, temperature=self.get_model().temperature_in)
            out = linear2(target_prob).view(n, t, -1)
            return out, target_prob.view(n, t, -1), out.detach().clone()
        ar_image_features = self._project(image_features)
        return ar_image_features, image_features.detach()

    def encode_imagesembed(self, image_features, return_prob=False):
        return self._project(image_features), image_features.detach()

    def _project(self, feats):
        F.params_ready(None)                                 # first read of trainable parameters in a forward pass
        proj = self.get_model().mm_projector
        n, t, c = feats.shape
        y = proj(feats.reshape(n * t, c).to(BF16).contiguous())
        return y.view(n, t, -1)

    # ------------------------------------------------------------------ A5
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, image_sizes=None, image_embeds=None, use_vision=True):
        vision_tower = self.get_vision_tower()
        if image_embeds is None:
            if vision_tower is None or images is None or input_ids.shape[1] == 1 or not use_vision:
                if not use_vision:
                    input_ids = input_ids[input_ids != -200]
                return input_ids, position_ids, attention_mask, past_key_values, None, labels, None, None
            if type(images) is list or images.ndim == 5:
                # the reference's list / 5-D branch is dead and broken (encode_images returns a tuple there and
                # target_features is never defined: metamorph_arch.py:192-239); refuse instead of reproducing it
                raise NotImplementedError("list / 5-D `images` (anyres) is not supported; pass a [N,3,H,W] tensor")
            image_features, target_features = self.encode_images(images)
        else:
            image_features, target_features = self.encode_imagesembed(image_embeds)

        cfg = self.config
        N, T, h = image_features.shape
        dev = image_features.device
        # The integer inputs on the host: CPU tensors as they are, device tensors through the mirror their mover registered
        # (metamorph_amd.hostmirror: MetaMorphTrainer / bench.py do) -- no device -> host copy, no synchronisation; unknown device tensors
        # fall back to ONE host copy (the reference syncs once per sample instead)
        ids_h, lab_h, msk_h = host_array(input_ids), host_array(labels), host_array(attention_mask)
        plan = build_splice_plan(ids_h, lab_h, msk_h, N, T, getattr(cfg, "tokenizer_model_max_length", None),
                                 getattr(cfg, "tokenizer_padding_side", "right"),
                                 getattr(cfg, "image_start_id", DEFAULT_IMAGE_START_ID),
                                 vocab_size=self.get_model().embed_tokens.weight.shape[0])
        keep_np = plan.target_keep.astype(np.int32) if 0 < plan.target_keep.shape[0] != N else None
        mask_np = None if attention_mask is None else plan.attention_mask.astype(np.bool_ if attention_mask.dtype == torch.bool else np.int64)
        pd = upload_plan(plan, dev, extra=dict(labels=plan.labels, mask=mask_np, image_positions=plan.image_positions,
                                               position_ids=None if position_ids is None else plan.position_ids, keep=keep_np))
        emb = self.get_model().embed_tokens
        proj2d = image_features.reshape(N * T, h)
        if torch.is_grad_enabled() and (emb.weight.requires_grad or proj2d.requires_grad):
            x = F.SpliceFn.apply(emb.weight, proj2d, emb, pd)
        else:
            x = ops.splice_gather(emb.weight.data, proj2d.detach(), pd["src"], h)
        inputs_embeds = x.view(plan.B, plan.L, h)

        new_labels = pd.get("x_labels")                      # (all four came over in the plan's one upload)
        if attention_mask is None:
            new_mask = None
        else:
            new_mask = pd["x_mask"] if pd["x_mask"].dtype == attention_mask.dtype else pd["x_mask"].to(attention_mask.dtype)
        new_pos = pd.get("x_position_ids")
        image_positions = pd["x_image_positions"]
        if plan.target_keep.shape[0] != N:                  # keep only answer-image features (metamorph_arch.py:415-423)
            Na = int(plan.target_keep.shape[0])
            if Na == 0:
                target_features = target_features[:0]
            else:
                C = target_features.shape[-1]
                target_features = ops.rows_gather(target_features.reshape(N, T * C).contiguous(), pd["x_keep"]).view(Na, T, C)
        # remember the plan for llm_forward (same tensor object => same plan)
        self._mm_plan = (inputs_embeds, pd)
        return None, new_pos, new_mask, past_key_values, inputs_embeds, new_labels, image_positions, target_features

    # ------------------------------------------------------------------ tokenizer / embedding resize
    def initialize_vision_tokenizer(self, model_args, tokenizer):
        """Reference metamorph_arch.py:427-469."""
        if model_args.mm_use_im_patch_token:
            tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
        if model_args.mm_use_im_start_end:
            num_new = tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
            if num_new > 0:
                ie = self.get_input_embeddings().weight.data
                oe = self.get_output_embeddings().weight.data
                ie[-num_new:] = ie[:-num_new].float().mean(dim=0, keepdim=True).to(ie.dtype)
                oe[-num_new:] = oe[:-num_new].float().mean(dim=0, keepdim=True).to(oe.dtype)
            if model_args.tune_mm_mlp_adapter:
                for p in self.get_input_embeddings().parameters():
                    p.requires_grad = True
                for p in self.get_output_embeddings().parameters():
                    p.requires_grad = False
            if getattr(model_args, "pretrain_mm_mlp_adapter", None):
                w = torch.load(model_args.pretrain_mm_mlp_adapter, map_location="cpu")["model.embed_tokens.weight"]
                assert num_new == 2
                ie = self.get_input_embeddings().weight.data
                if ie.shape == w.shape:
                    ie[-num_new:] = w[-num_new:]
                elif w.shape[0] == num_new:
                    ie[-num_new:] = w
                else:
                    raise ValueError(f"Unexpected embed_tokens_weight shape. Pretrained: {w.shape}. Current: {ie.shape}. "
                                     f"Numer of new tokens: {num_new}.")
        elif model_args.mm_use_im_patch_token:
            if model_args.tune_mm_mlp_adapter:
                for p in self.get_input_embeddings().parameters():
                    p.requires_grad = False
                for p in self.get_output_embeddings().parameters():
                    p.requires_grad = False
