"""Checkpoint I/O in the layouts the reference's orchestration reads and writes.

* stage-1 adapter files: `mm_projector.bin` holding every parameter whose name contains one of the adapter keys
  (`mm_projector`, optionally `embed_tokens` / `embed_in` when `<image_start>/<image_end>` embeddings are trained), placed
  exactly where the reference puts them (reference train.py:163-166, 186-209 and metamorph_trainer.py:273-292): a folder named
  `checkpoint-<step>` saves to `<parent>/mm_projector/checkpoint-<step>.bin`, anything else to `<dir>/mm_projector.bin`;
  `MetaMorphMetaModel.initialize_vision_modules` loads them back through `pretrain_mm_mlp_adapter`;
* full models go through `save_pretrained` / `from_pretrained` (HF layout, state-dict keys unchanged);
* the ZeRO-2 optimizer state is sharded by rank AND by gradient segment; `consolidate_optimizer_state` gathers it into
  world-size-independent per-parameter fp32 tensors (the role of DeepSpeed's zero_to_fp32 for the reference's checkpoints) and
  `load_consolidated_optimizer_state` scatters such a file into an optimizer of any world size.

With ZeRO-2 every rank holds the complete bf16 parameters, so no gather is needed for the weights themselves.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Sequence, Tuple

import torch
import torch.distributed as dist


def get_mm_adapter_state(named_params: Iterable[Tuple[str, torch.Tensor]], keys_to_match: Sequence[str]) -> Dict[str, torch.Tensor]:
    """name -> CPU tensor for every parameter whose name contains one of `keys_to_match`."""
    return {k: t.detach().cpu() for k, t in named_params if any(key in k for key in keys_to_match)}


def adapter_keys(use_im_start_end: bool = False, trainer_checkpoint: bool = False):
    keys = ["mm_projector"] + (["vision_resampler"] if trainer_checkpoint else [])
    if use_im_start_end:
        keys += ["embed_tokens", "embed_in"]
    return keys


def save_mm_adapter(model, output_dir: str, use_im_start_end: bool = False, is_main_process: bool = True) -> str:
    """The reference's `safe_save_model_for_hf_trainer` branch for `tune_mm_mlp_adapter` runs.  Returns the file written."""
    state = get_mm_adapter_state(model.named_parameters(), adapter_keys(use_im_start_end))
    model.config.save_pretrained(output_dir)
    folder = output_dir.rstrip("/").split("/")[-1]
    if folder.startswith("checkpoint-"):
        target_dir = os.path.join(os.path.dirname(output_dir.rstrip("/")), "mm_projector")
        path = os.path.join(target_dir, f"{folder}.bin")
    else:
        target_dir = output_dir
        path = os.path.join(output_dir, "mm_projector.bin")
    if is_main_process:
        os.makedirs(target_dir, exist_ok=True)
        torch.save(state, path)
    return path


def save_trainer_adapter_checkpoint(model, run_dir: str, global_step: int, use_im_start_end: bool = False,
                                    is_main_process: bool = True) -> str:
    """The reference's `MetaMorphTrainer._save_checkpoint` for adapter-only runs: <run_dir>/checkpoint-<step>/mm_projector.bin."""
    out = os.path.join(run_dir, f"checkpoint-{global_step}")
    state = get_mm_adapter_state(model.named_parameters(), adapter_keys(use_im_start_end, trainer_checkpoint=True))
    path = os.path.join(out, "mm_projector.bin")
    if is_main_process:
        os.makedirs(out, exist_ok=True)
        model.config.save_pretrained(out)
        torch.save(state, path)
    return path


def safe_save_model(model, output_dir: str, tune_mm_mlp_adapter: bool = False, use_im_start_end: bool = False,
                    is_main_process: bool = True, optimizer=None):
    """Adapter-only runs write the adapter file; everything else the full HF checkpoint (rank 0 only: ZeRO-2 ranks hold
    identical complete parameters).  Pass the Zero2AdamW so that a pending asynchronous update is waited for first."""
    if optimizer is not None:
        optimizer.synchronize()
    if tune_mm_mlp_adapter:
        return save_mm_adapter(model, output_dir, use_im_start_end, is_main_process)
    if is_main_process:
        model.save_pretrained(output_dir, state_dict={k: v.detach().cpu() for k, v in model.state_dict().items()})
    return output_dir


# ------------------------------------------------------------------------------------------------ ZeRO-2 optimizer state
def _param_names(model, opt):
    by_id = {id(p): n for n, p in model.named_parameters()}
    return [by_id[id(p)] for p in opt.params]


def consolidate_optimizer_state(opt, model, dst: int = 0):
    opt.wait_all()
    return _consolidate(opt, model, dst)


def _consolidate(opt, model, dst):
    """All ranks call this.  Returns on rank `dst` {"step", "param_groups", "state": {name: {"master", "exp_avg", "exp_avg_sq"}}}
    with full fp32 tensors in each parameter's own shape (CPU), independent of world size and segmentation; None elsewhere."""
    names = _param_names(model, opt)
    full = {}
    for key in ("master", "exp_avg", "exp_avg_sq"):
        shard = getattr(opt, key)
        flat = torch.zeros(opt.padded, dtype=torch.float32, device=shard.device)
        for sg in opt.segs:                                  # this rank's slice of every segment back to its flat position
            lo = sg["lo"] + opt.rank * sg["m"]
            flat[lo:lo + sg["m"]] = shard[sg["so"]:sg["so"] + sg["m"]]
        if opt.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=opt.pg)      # disjoint slices: the sum is the concatenation
        full[key] = flat.cpu()
    if opt.rank != dst:
        return None
    state = {}
    for name, p, off in zip(names, opt.params, opt.offsets):
        state[name] = {k: full[k][off:off + p.numel()].view(p.shape).clone() for k in full}
    return {"step": opt._step, "state": state,
            "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in opt.param_groups]}


def load_consolidated_optimizer_state(opt, model, ckpt):
    """Scatter a `consolidate_optimizer_state` result into this optimizer's shards (any world size / segmentation)."""
    names = _param_names(model, opt)
    missing = [n for n in names if n not in ckpt["state"]]
    if missing:
        raise KeyError(f"optimizer checkpoint lacks {len(missing)} parameters, e.g. {missing[:3]}")
    for key in ("master", "exp_avg", "exp_avg_sq"):
        shard = getattr(opt, key)
        flat = torch.zeros(opt.padded, dtype=torch.float32)
        for name, p, off in zip(names, opt.params, opt.offsets):
            t = ckpt["state"][name][key]
            if tuple(t.shape) != tuple(p.shape):
                raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != parameter shape {tuple(p.shape)}")
            flat[off:off + p.numel()] = t.reshape(-1).float()
        for sg in opt.segs:
            lo = sg["lo"] + opt.rank * sg["m"]
            shard[sg["so"]:sg["so"] + sg["m"]].copy_(flat[lo:lo + sg["m"]])
    opt._step = int(ckpt["step"])
    for g, s in zip(opt.param_groups, ckpt["param_groups"]):
        g.update(s)
    # the bf16 parameters follow the fp32 master copy
    for sg in opt.segs:
        sg["my_param"].copy_(opt.master[sg["so"]:sg["so"] + sg["m"]].to(sg["my_param"].dtype))
    opt._all_gather_params()
    if opt.master.is_cuda:                                   # parameters changed behind torch's version counters
        from . import functional as F
        F.bump_param_generation()
