"""Zero3AdamW on the HIP model (single MI355X): decoder-layer parameters live only as shards, are gathered per layer in forward /
recompute / backward through the functional hooks, gradients leave through rotating slots -- and training must give what Zero2AdamW
gives.  BASELINE configs[4] (reference scripts/zero3.json).  Needs an MI355X:  pytest -m gpu"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle.ref_model import init_state_dict  # noqa: E402
from test_model_gpu import T, hip_model, tiny_cfg  # noqa: E402


def _batch():
    g = np.load(os.path.join(GOLDEN, "e2e_multi_frame_T4_ar1_bf16.npz"))
    return g, dict(input_ids=T(g["input_ids"]).cuda(), attention_mask=T(g["attention_mask"]).cuda(), labels=T(g["labels"]).cuda(),
                   images=T(g["images"]).cuda().bfloat16())


def _train(stage, steps=3, accum=1, ckpt=False, clip=0.0, layers=3, param_slots=2):
    from metamorph_amd import functional as F
    from metamorph_amd.zero2 import Zero2AdamW, tag_segments
    from metamorph_amd.zero3 import Zero3AdamW
    g, args = _batch()
    cfg = tiny_cfg(num_image_tokens=4, num_hidden_layers=layers)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    model.train()
    if ckpt:
        model.gradient_checkpointing_enable()
    tag_segments(model)
    params = [p for p in model.parameters() if p.requires_grad]
    if stage == 3:
        opt = Zero3AdamW(params, lr=1e-3, max_grad_norm=clip, param_slots=param_slots, grad_slots=1, min_shard_numel=1).enable_hooks()
    else:
        opt = Zero2AdamW(params, lr=1e-3, max_grad_norm=clip).enable_overlap()
    losses = []
    try:
        for _ in range(steps):
            opt.zero_grad()
            for _ in range(accum):
                out = model(**args)
                opt.arm_overlap()
                (out.loss / accum).backward()
            opt.step()
            losses.append(float(out.loss.detach()))
        opt.synchronize()
        if stage == 3:
            # the layers hold no storage between steps; evaluation gathers them again through the same hook
            assert all(p.data.numel() == 0 for l in model.get_model().layers for p in l.parameters())
            model.eval()
            with torch.no_grad():
                ev = float(model(**args).loss)
            full = opt.gather_full_parameters()
            flat = torch.cat([(full[p] if p in full else p.data).reshape(-1).float() for p in params])
        else:
            model.eval()
            with torch.no_grad():
                ev = float(model(**args).loss)
            flat = torch.cat([p.data.reshape(-1).float() for p in params])
    finally:
        F.set_layer_grad_hook(None)
        F.set_param_ready_hook(None)
    return losses, ev, flat, opt


@pytest.mark.parametrize("ckpt", [False, True])
def test_zero3_equals_zero2_bit_for_bit(ckpt):
    """No clipping, no accumulation: the same kernels see the same bytes
    whether the parameters come from the flat buffer or from a gathered slot => identical losses and parameters.  With recompute
    (`gradient_checkpointing`, the 70B recipe) the layer is gathered three times per step: forward, recompute, backward."""
    l2, e2, p2, _ = _train(2, ckpt=ckpt)
    l3, e3, p3, opt = _train(3, ckpt=ckpt)
    # (the scalar loss is an fp32 atomicAdd over rows: equal to rounding, not bit for bit; the gradients do not depend on it)
    assert all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(l2, l3)) and abs(e2 - e3) <= 1e-6 * abs(e2), (l2, l3, e2, e3)
    assert torch.equal(p2, p3)
    rep = opt.memory_report()
    assert rep["param_slots"] > 0 and rep["param_shards"] > 0 and l3[-1] < l3[0]


def test_zero3_accumulation_and_clipping_close_to_zero2():
    """Gradient accumulation folds every micro-step's reduce-scattered slice into the gradient shard (one more bf16 rounding than
    ZeRO-2's in-epilogue accumulation) and clipping uses the sharded norm: same training curve to bf16 noise."""
    l2, e2, p2, _ = _train(2, accum=2, clip=1.0)
    l3, e3, p3, _ = _train(3, accum=2, clip=1.0, param_slots=3)
    for a, b in zip(l2, l3):
        assert abs(a - b) <= 2e-3 * abs(a), (l2, l3)
    assert abs(e2 - e3) <= 2e-3 * abs(e2)
    assert float((p2 - p3).norm() / p2.norm()) <= 2e-3


def test_zero3_rccl_call_pattern_single_rank():
    """All-gather into a slot at the shard's own offset (in place), asynchronous prefetch of the next layer, reduce-scatter out of the
    gradient slot with a deferred accumulate -- through real RCCL with a one-rank process group."""
    import torch.distributed as dist
    base, ebase, pbase, _ = _train(3)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    _old_mode = __import__("metamorph_amd.zero2", fromlist=["x"]).set_collective_mode(force_collectives=True)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        losses, ev, flat, opt = _train(3)
        assert opt._coll and opt.world == 1
        assert all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(losses, base)) and abs(ev - ebase) <= 1e-6 * abs(ev)
        assert torch.equal(flat, pbase)
    finally:
        dist.destroy_process_group()
        __import__("metamorph_amd.zero2", fromlist=["x"]).set_collective_mode(**_old_mode)
