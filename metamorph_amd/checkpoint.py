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


# ------------------------------------------------------------------------------------------------ DeepSpeed's own ZeRO-2 shards
# The reference trains under DeepSpeed (scripts/zero2.json); an HF Trainer checkpoint of such a run holds, next to the consolidated
# weights, DeepSpeed's per-rank resume files (HF `Trainer._save_checkpoint` -> `deepspeed_engine.save_checkpoint`, reached from the
# reference at train.py:210-213 / through `trainer.train(resume_from_checkpoint=True)`, train.py:1592-1599):
#
#   checkpoint-N/latest                                              text: "global_stepN"
#   checkpoint-N/global_stepN/mp_rank_00_model_states.pt             {"module": 16-bit state dict, "param_shapes": [OrderedDict name -> shape per
#                                                                     parameter group], "ds_version", ...}
#   checkpoint-N/global_stepN/[bf16_]zero_pp_rank_{r}_mp_rank_00_optim_states.pt
#                                                                    {"optimizer_state_dict": {"zero_stage": 2, "partition_count": W,
#                                                                     "single_partition_of_fp32_groups": [flat fp32 slice per group],
#                                                                     "base_optimizer_state": {"state": {g: {"exp_avg", "exp_avg_sq", "step"}},
#                                                                                              "param_groups": [...]}}}
#
# A group's fp32 master vector is the concatenation of the ranks' slices: its parameters back to back in `param_shapes[g]` order, padded
# to a multiple of 2 x world size (DeepSpeed's `zero_to_fp32.py`, pinned 0.15.1 in the reference's pyproject.toml:16: `zero2_align`); the
# Adam moments are partitioned the same way.  DeepSpeed is NOT installed in this image, so no file of its making could be recorded: the
# layout above is restated from that script -- **parity unpinned** (tests/test_host_logic.py round-trips it through a writer of the same
# layout; a real checkpoint has not been read).

def _ds_step_dir(checkpoint_dir: str) -> str:
    latest = os.path.join(checkpoint_dir, "latest")
    if os.path.isfile(latest):
        return os.path.join(checkpoint_dir, open(latest).read().strip())
    steps = sorted(d for d in os.listdir(checkpoint_dir) if d.startswith("global_step"))
    if not steps:
        raise FileNotFoundError(f"{checkpoint_dir}: no `latest` file and no global_step* directory (not a DeepSpeed checkpoint)")
    return os.path.join(checkpoint_dir, steps[-1])


def read_deepspeed_zero2_checkpoint(checkpoint_dir: str):
    """DeepSpeed ZeRO-2 per-rank shards -> the world-size-independent form `load_consolidated_optimizer_state` takes:
    {"step", "param_groups", "state": {name: {"master", "exp_avg", "exp_avg_sq"}}} (fp32 CPU tensors in each parameter's shape)."""
    import re
    step_dir = _ds_step_dir(checkpoint_dir)
    model_states = torch.load(os.path.join(step_dir, "mp_rank_00_model_states.pt"), map_location="cpu", weights_only=False)
    shapes = model_states["param_shapes"]
    files = {}
    for f in os.listdir(step_dir):
        m = re.fullmatch(r"(?:bf16_)?zero_pp_rank_(\d+)_mp_rank_00_optim_states\.pt", f)
        if m:
            files[int(m.group(1))] = os.path.join(step_dir, f)
    if not files or sorted(files) != list(range(len(files))):
        raise FileNotFoundError(f"{step_dir}: optimizer shards of ranks {sorted(files)} found; need 0 .. W-1")
    osd = [torch.load(files[r], map_location="cpu", weights_only=False)["optimizer_state_dict"] for r in range(len(files))]
    stage = osd[0].get("zero_stage", 2)
    if int(stage) > 2:
        raise NotImplementedError(f"zero_stage {stage}: ZeRO-3 shards partition every parameter separately (use the gathered 16-bit weights instead)")
    world = osd[0]["partition_count"]
    world = int(max(world) if isinstance(world, (list, tuple)) else world)
    if world != len(files):
        raise ValueError(f"partition_count {world} but {len(files)} shard files")
    state, step = {}, 0
    for g, group_shapes in enumerate(shapes):
        def merged(pick):
            return torch.cat([pick(osd[r]).reshape(-1).float() for r in range(world)])
        vec = {"master": merged(lambda o: o["single_partition_of_fp32_groups"][g]),
               "exp_avg": merged(lambda o: o["base_optimizer_state"]["state"][g]["exp_avg"]),
               "exp_avg_sq": merged(lambda o: o["base_optimizer_state"]["state"][g]["exp_avg_sq"])}
        st = osd[0]["base_optimizer_state"]["state"][g].get("step", 0)
        step = max(step, int(st.item() if isinstance(st, torch.Tensor) else st))
        off = 0
        for name, shape in group_shapes.items():
            n = 1
            for d in shape:
                n *= int(d)
            if off + n > vec["master"].numel():
                raise ValueError(f"group {g}: parameter {name} ends at {off + n}, the merged partitions hold {vec['master'].numel()} values")
            state[name] = {k: v[off:off + n].view(tuple(shape)).clone() for k, v in vec.items()}
            off += n
        align = 2 * world
        if (off + align - 1) // align * align != vec["master"].numel():
            raise ValueError(f"group {g}: {off} parameter values aligned to {align} != {vec['master'].numel()} values in the merged partitions")
    groups = [{k: v for k, v in pg.items() if k != "params"} for pg in osd[0]["base_optimizer_state"]["param_groups"]]
    return {"step": step, "state": state, "param_groups": groups, "module": model_states.get("module")}


def write_deepspeed_zero2_layout(consolidated, group_names, checkpoint_dir: str, world: int, tag: str = "global_step1", bf16: bool = True,
                                 module=None):
    """The inverse of `read_deepspeed_zero2_checkpoint` (used by its test and for handing a run BACK to a DeepSpeed stack): the consolidated
    state as W per-rank shard files + the model-states file.  group_names: one list of parameter names per optimizer parameter group."""
    from collections import OrderedDict
    step_dir = os.path.join(checkpoint_dir, tag)
    os.makedirs(step_dir, exist_ok=True)
    align = 2 * world
    flats, shapes = [], []
    for names in group_names:
        shapes.append(OrderedDict((n, torch.Size(consolidated["state"][n]["master"].shape)) for n in names))
        flat = {}
        for k in ("master", "exp_avg", "exp_avg_sq"):
            v = torch.cat([consolidated["state"][n][k].reshape(-1).float() for n in names])
            pad = (v.numel() + align - 1) // align * align - v.numel()
            flat[k] = torch.cat([v, v.new_zeros(pad)])
        flats.append(flat)
    torch.save({"module": module or {}, "param_shapes": shapes, "buffer_names": [], "shared_params": {}, "ds_version": "0.15.1"},
               os.path.join(step_dir, "mp_rank_00_model_states.pt"))
    for r in range(world):
        part = lambda v: v.view(world, -1)[r].clone()
        osd = {"zero_stage": 2, "partition_count": world, "loss_scaler": None, "dynamic_loss_scale": False, "overflow": False, "clip_grad": 1.0,
               "single_partition_of_fp32_groups": [part(f["master"]) for f in flats],
               "base_optimizer_state": {"state": {g: {"exp_avg": part(f["exp_avg"]), "exp_avg_sq": part(f["exp_avg_sq"]), "step": consolidated["step"]}
                                                  for g, f in enumerate(flats)},
                                        "param_groups": [dict(pg, params=[g]) for g, pg in enumerate(consolidated["param_groups"])]},
               "ds_version": "0.15.1"}
        torch.save({"optimizer_state_dict": osd, "ds_version": "0.15.1"},
                   os.path.join(step_dir, f"{'bf16_' if bf16 else ''}zero_pp_rank_{r}_mp_rank_00_optim_states.pt"))
    with open(os.path.join(checkpoint_dir, "latest"), "w") as f:
        f.write(tag)
    return step_dir
