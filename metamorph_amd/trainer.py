"""`MetaMorphTrainer`: the binding between HF `Trainer` and the MI355X hot path (reference metamorph/train/metamorph_trainer.py:138-298).

The reference's trainer subclass exists to (a) build the optimizer parameter groups (`mm_projector_lr` / `vision_lr`, weight decay off
for biases and norm weights) and (b) save only the adapter in stage 1; gradient exchange and optimizer-state sharding are delegated to
DeepSpeed through `--deepspeed scripts/zero2.json`.  Here the same subclass name keeps (a) and (b) and replaces the DeepSpeed
delegation with `Zero2AdamW` (metamorph_amd/zero2.py):

  * `create_optimizer` builds the reference's groups and hands them to `Zero2AdamW` (flat bf16 parameter / gradient buffers, per-layer
    reduce-scatter over RCCL started from inside the backward pass, sharded fused AdamW, all-gather);
  * the model is NOT wrapped in DistributedDataParallel: the decoder's autograd nodes write weight gradients straight into the flat
    gradient buffer and return None to autograd, so DDP's reducer hooks would never fire (and DDP would average the few autograd-routed
    gradients a second time).  `Zero2AdamW` is the only gradient exchange;
  * gradient clipping moves into the optimizer (`max_grad_norm` is the norm of the MEAN gradient over ranks, computed from the sharded
    reduce-scattered buffer; HF's own `clip_grad_norm_` would see un-reduced local gradients);
  * `training_step` arms the overlapped reduce-scatter on the last micro-step of an accumulation window;
  * `gradient_checkpointing=True` (every reference launch script) maps onto per-layer recompute in `DecoderLayerFn`;
  * optimizer checkpoints: HF writes `optimizer.state_dict()` from rank 0 only, which under a sharded optimizer is one rank's slice.
    `_save_optimizer_and_scheduler` / `_load_optimizer_and_scheduler` write and read the whole state instead: ZeRO-2 as ONE
    world-size-independent file (`checkpoint.consolidate_optimizer_state`: fp32 master / moments per parameter name), ZeRO-3 as one
    shard file per rank (the reference's DeepSpeed checkpoints are per-rank `zero_pp_rank_*` files as well).
"""
from __future__ import annotations

import os
import warnings

import torch
from transformers import Trainer

from .checkpoint import consolidate_optimizer_state, load_consolidated_optimizer_state, save_trainer_adapter_checkpoint
from . import functional as F
from . import hostmirror
from .zero2 import Zero2AdamW, tag_segments
from .zero3 import Zero3AdamW


def _decay_parameter_names(model):
    """Names that receive weight decay: everything except biases and normalisation weights (HF `get_parameter_names(model,
    ALL_LAYERNORM_LAYERS)` minus "bias", reference metamorph_trainer.py:170-171)."""
    import torch.nn as nn
    from .model.modules import HipLayerNorm, HipRMSNorm
    norm_types = (nn.LayerNorm, HipLayerNorm, HipRMSNorm)
    skip = set()
    for mname, mod in model.named_modules():
        if isinstance(mod, norm_types):
            for pname, _ in mod.named_parameters(recurse=False):
                skip.add(f"{mname}.{pname}" if mname else pname)
    return [n for n, _ in model.named_parameters() if n not in skip and "bias" not in n]


def optimizer_grouped_parameters(model, weight_decay, mm_projector_lr=None, vision_lr=None):
    """The reference's four-way split (decay x {base lr, special lr}), reference metamorph_trainer.py:172-245."""
    decay = set(_decay_parameter_names(model))
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    special, special_lr = set(), None
    if mm_projector_lr is not None:
        special, special_lr = {n for n, _ in named if "mm_projector" in n}, mm_projector_lr
    elif vision_lr is not None:
        special, special_lr = {n for n, _ in named if "vision_tower" in n}, vision_lr
    groups = [
        {"params": [p for n, p in named if n in decay and n not in special], "weight_decay": weight_decay},
        {"params": [p for n, p in named if n not in decay and n not in special], "weight_decay": 0.0},
    ]
    if special:
        groups += [
            {"params": [p for n, p in named if n in decay and n in special], "weight_decay": weight_decay, "lr": special_lr},
            {"params": [p for n, p in named if n not in decay and n in special], "weight_decay": 0.0, "lr": special_lr},
        ]
    return [g for g in groups if g["params"]]


class MetaMorphTrainer(Trainer):
    """Drop-in for the reference's `MetaMorphTrainer`; `zero2_kwargs` are forwarded to `Zero2AdamW` (tests inject CPU shard kernels)."""

    def __init__(self, *args, zero2_kwargs=None, zero_stage=2, **kwargs):
        # zero_stage: 2 = Zero2AdamW (reference scripts/zero2.json), 3 = Zero3AdamW (scripts/zero3.json: decoder-layer parameters sharded)
        if zero_stage not in (2, 3):
            raise ValueError("zero_stage must be 2 or 3")
        self._zero_stage = zero_stage
        self._zero2_kwargs = dict(zero2_kwargs or {})
        self._mm_max_grad_norm = None
        super().__init__(*args, **kwargs)
        # clipping happens inside Zero2AdamW on the reduced gradient (see the module docstring)
        self._mm_max_grad_norm = self.args.max_grad_norm
        self.args.max_grad_norm = 0.0

    # ------------------------------------------------------------------ no DDP wrapper
    def create_accelerator_and_postprocess(self):
        super().create_accelerator_and_postprocess()
        acc = self.accelerator
        inner = acc.prepare_model

        def prepare_model(model, device_placement=None, evaluation_mode=False):
            if getattr(model, "_mm355_no_ddp", False) or hasattr(model, "get_model"):
                # Zero2AdamW exchanges the gradients; a DDP reducer would wait for hooks that never fire
                if model not in acc._models:
                    acc._models.append(model)
                return model
            return inner(model, device_placement=device_placement, evaluation_mode=evaluation_mode)

        acc.prepare_model = prepare_model

        # Batches stay on the HOST until `_prepare_inputs`: accelerate's prepared DataLoader would move them to the device itself
        # (device_placement defaults to True), `_prepare_inputs` would then see device tensors only, no host original could be registered as
        # a mirror, and every step's splice plan would pull ids / labels / mask back with a synchronising `.cpu()` (hostmirror.py).
        inner_dl = acc.prepare_data_loader

        def prepare_data_loader(data_loader, device_placement=None, slice_fn_for_dispatch=None):
            return inner_dl(data_loader, device_placement=False, slice_fn_for_dispatch=slice_fn_for_dispatch)

        acc.prepare_data_loader = prepare_data_loader

    # ------------------------------------------------------------------ optimizer
    def create_optimizer(self):
        if self.optimizer is not None:
            return self.optimizer
        a = self.args
        tag_segments(self.model)                                     # one reduce-scatter segment per decoder layer
        groups = optimizer_grouped_parameters(self.model, a.weight_decay, getattr(a, "mm_projector_lr", None), getattr(a, "vision_lr", None))
        max_norm = self._mm_max_grad_norm if self._mm_max_grad_norm is not None else a.max_grad_norm
        cls = Zero3AdamW if self._zero_stage == 3 else Zero2AdamW
        self.optimizer = cls(groups, lr=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2), eps=a.adam_epsilon,
                             weight_decay=a.weight_decay, max_grad_norm=max_norm or 0.0, **self._zero2_kwargs)
        self.optimizer.enable_overlap()
        # gradient accumulation: the transposed weight copies of the input-gradient GEMMs live until the optimizer rewrites the parameters
        # (functional.transposed_weight: keyed on the parameter generation Zero2AdamW bumps).  Not under ZeRO-3: its gathered layers share
        # rotating slots, i.e. one address holds different layers within a step.
        F.set_variant("wt_cache", a.gradient_accumulation_steps > 1 and self._zero_stage != 3)
        return self.optimizer

    def train(self, *args, **kwargs):
        try:
            return super().train(*args, **kwargs)
        finally:                                                     # the transposed-weight copies belong to this run's optimizer steps
            F.set_variant("wt_cache", False)
            F.drop_transposed_weights()

    def _zero2(self):
        opt = self.optimizer
        while opt is not None and not isinstance(opt, (Zero2AdamW, Zero3AdamW)):
            opt = getattr(opt, "optimizer", None)                    # accelerate's AcceleratedOptimizer wrapper
        return opt

    def _prepare_inputs(self, inputs):
        """HF moves the collator's batch to the device here; the integer tensors the splice plan is built from (reference batch contract,
        train.py:1258-1284) keep their host originals registered as mirrors (metamorph_amd.hostmirror), so the model's forward neither copies
        them back nor synchronises with the device."""
        out = super()._prepare_inputs(inputs)
        if isinstance(inputs, dict) and isinstance(out, dict):
            for k in ("input_ids", "labels", "attention_mask"):
                src, dst = inputs.get(k), out.get(k)
                if isinstance(src, torch.Tensor) and isinstance(dst, torch.Tensor) and src.device.type == "cpu" and dst.device.type != "cpu" \
                        and src.dtype == dst.dtype and src.shape == dst.shape:
                    hostmirror.attach(dst, src)
        return out

    def training_step(self, model, inputs, num_items_in_batch=None):
        z = self._zero2()
        if z is not None and self.accelerator.sync_gradients:
            z.arm_overlap()                                          # last micro-step: these gradients are final
        return super().training_step(model, inputs, num_items_in_batch)

    # ------------------------------------------------------------------ stage-1 adapter checkpoints (reference :273-298)
    def _save_checkpoint(self, model, trial, metrics=None):
        if getattr(self.args, "tune_mm_mlp_adapter", False):
            save_trainer_adapter_checkpoint(self.model, self._get_output_dir(trial=trial), self.state.global_step,
                                            use_im_start_end=getattr(self.args, "use_im_start_end", False),
                                            is_main_process=self.args.local_rank in (0, -1))
            return
        z = self._zero2()
        if z is not None:
            z.synchronize()
        super()._save_checkpoint(model, trial)

    # ------------------------------------------------------------------ optimizer state: every rank's shard, not rank 0's
    _OPT, _SCHED = "optimizer.pt", "scheduler.pt"

    def _save_optimizer_and_scheduler(self, output_dir):
        z = self._zero2()
        if z is None:
            return super()._save_optimizer_and_scheduler(output_dir)
        os.makedirs(output_dir, exist_ok=True)
        if isinstance(z, Zero3AdamW):
            torch.save(z.state_dict(), os.path.join(output_dir, f"zero3_rank{z.rank}-of-{z.world}-{self._OPT}"))
            marker = {"zero_stage": 3, "world": z.world}                 # what HF's resume logic looks for
        else:
            marker = consolidate_optimizer_state(z, self.model)          # collective: every rank takes part, rank 0 gets the result
        if self.args.should_save:
            torch.save(marker, os.path.join(output_dir, self._OPT))
            with warnings.catch_warnings(record=True):
                torch.save(self.lr_scheduler.state_dict(), os.path.join(output_dir, self._SCHED))

    def _load_optimizer_and_scheduler(self, checkpoint):
        z = self._zero2()
        if checkpoint is None or z is None:
            return super()._load_optimizer_and_scheduler(checkpoint)
        path = os.path.join(checkpoint, self._OPT)
        if not os.path.isfile(path):
            # a checkpoint written by the reference's own stack: DeepSpeed's per-rank ZeRO-2 shards (global_step*/[bf16_]zero_pp_rank_*): merged into
            # the world-size-independent form and scattered into this optimizer (checkpoint.read_deepspeed_zero2_checkpoint; layout unpinned)
            if not isinstance(z, Zero3AdamW) and (os.path.isfile(os.path.join(checkpoint, "latest")) or any(
                    d.startswith("global_step") for d in os.listdir(checkpoint))):
                from .checkpoint import read_deepspeed_zero2_checkpoint
                ds = read_deepspeed_zero2_checkpoint(checkpoint)
                ds["param_groups"] = ds["param_groups"][:len(z.param_groups)] if len(ds["param_groups"]) >= len(z.param_groups) else [{}] * len(z.param_groups)
                load_consolidated_optimizer_state(z, self.model, ds)
            return
        if isinstance(z, Zero3AdamW):
            shard = os.path.join(checkpoint, f"zero3_rank{z.rank}-of-{z.world}-{self._OPT}")
            if not os.path.isfile(shard):
                raise FileNotFoundError(f"{shard}: ZeRO-3 optimizer shards are per rank; resume with the world size that wrote them")
            z.load_state_dict(torch.load(shard, map_location=z.master.device, weights_only=True))
        else:
            load_consolidated_optimizer_state(z, self.model, torch.load(path, map_location="cpu", weights_only=True))
        sched = os.path.join(checkpoint, self._SCHED)
        if os.path.isfile(sched):
            with warnings.catch_warnings(record=True):
                self.lr_scheduler.load_state_dict(torch.load(sched, map_location="cpu", weights_only=True))

    def save_model(self, output_dir=None, _internal_call=False):
        """HF's `save_model` calls `_save` on the `should_save` rank only; under ZeRO-3 the model's decoder layers exist as per-rank
        shards, so the gather behind `_save` is a collective EVERY rank must enter (HF does the same in its DeepSpeed / FSDP branches,
        transformers Trainer.save_model).  All ranks gather layer by layer; only the saving rank keeps (host copies of) the tensors."""
        z = self._zero2()
        if not isinstance(z, Zero3AdamW) or getattr(self.args, "tune_mm_mlp_adapter", False):
            return super().save_model(output_dir, _internal_call=_internal_call)
        if output_dir is None:
            output_dir = self.args.output_dir
        z.synchronize()
        state_dict = z.full_state_dict(self.model, device="cpu", keep=bool(self.args.should_save))
        if self.args.should_save:
            self._save(output_dir, state_dict=state_dict)
        if self.args.push_to_hub and not _internal_call:
            self.push_to_hub(commit_message="Model save", revision=getattr(self.args, "hub_revision", None))

    def _save(self, output_dir=None, state_dict=None):
        if getattr(self.args, "tune_mm_mlp_adapter", False):
            return
        z = self._zero2()
        if state_dict is None and isinstance(z, Zero3AdamW):
            if z.world > 1:
                raise RuntimeError("_save under ZeRO-3 needs the gathered state dict: call save_model() (a collective every rank enters)")
            state_dict = z.full_state_dict(self.model, device="cpu")
        elif z is not None:
            z.synchronize()                                          # an asynchronous update / all-gather may still be writing the parameters
        super()._save(output_dir, state_dict)
