"""ZeRO-2 partition / collective logic on CPU: world_size 2 over gloo must reproduce a single-rank AdamW step on
the mean gradient (the reference's DeepSpeed semantics: gradients averaged over data-parallel ranks, global-norm
clipping, one AdamW update; scripts/zero2.json + HF Trainer).  The shard-update arithmetic is injected from the
oracle here because the HIP kernel cannot run on CPU; on the GPU box the same class runs with the kernel
(tests/test_zero2_gpu.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_ops as R


def _oracle_update(p32, m, v, g, p_out, lr, b1, b2, eps, wd, step, scale_dev):
    R.adamw_step(p32, g, m, v, step, lr, b1, b2, eps, wd, grad_scale=float(scale_dev))
    p_out.copy_(p32.to(p_out.dtype))


def _oracle_sumsq(x, out):
    out += (x.float() ** 2).sum()


def _oracle_clip(sumsq, max_norm, pre, out):
    c = 1.0 if max_norm <= 0 else min(1.0, max_norm / (float(sumsq) ** 0.5 + 1e-6))
    out[0] = c * pre


def _make_params(dtype):
    g = torch.Generator().manual_seed(0)
    shapes = [(8, 16), (4, 16), (4, 16), (16, 16), (37,), (5, 3)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dtype)) for s in shapes]


def _grads_for(rank, step, params):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return [torch.randn(p.shape, generator=g).to(p.dtype) * 3.0 for p in params]


# segment keys of _make_params' six tensors: two "decoder layers" ([0,1,2] fused q/k/v-like block, [3,4]) and a free tail
_SEGMENTS = [("layer", 0), ("layer", 0), ("layer", 0), ("layer", 1), ("layer", 1), None]


def _worker(rank, world, port, dtype, steps, tmp, overlap=False, tensor_coll=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if tensor_coll:     # the RCCL form of the exchange (in-place reduce_scatter_tensor / all_gather_into_tensor on buffer slices), run on gloo
        __import__("metamorph_amd.zero2", fromlist=["x"]).set_collective_mode(tensor_collectives=True)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metamorph_amd.zero2 import Zero2AdamW
        params = _make_params(dtype)
        for p, key in zip(params, _SEGMENTS):
            if key is not None:
                p._mm_segment = key
        opt = Zero2AdamW(params, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, max_grad_norm=1.0,
                         shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip, overlap=overlap)
        assert opt.world == world and opt.shard * world == opt.padded and len(opt.segs) == 3 and opt._tensor_coll == bool(tensor_coll)
        assert all(sg["n"] % (world * 256) == 0 for sg in opt.segs)
        # fused blocks stay contiguous in the flat buffer (q/k/v -> one GEMM operand)
        assert params[1].data_ptr() == params[0].data_ptr() + params[0].numel() * params[0].element_size()
        for s in range(1, steps + 1):
            if overlap:
                opt.arm_overlap()
            grads = _grads_for(rank, s, params)
            # backward order: the tail first, then layer 1, then layer 0 -- each layer announces itself when done
            for idx in (5, 4, 3, 2, 1, 0):
                if not (s == 2 and idx == 5):        # a parameter without gradient this step contributes zeros
                    params[idx]._mm_grad_buf.copy_(grads[idx])   # what the backward kernels do
                    params[idx].grad = params[idx]._mm_grad_buf
                if idx == 3:
                    opt.notify_segment_ready(("layer", 1))
                    assert (1 in opt._pending) == (overlap and world > 1)
                if idx == 0:
                    opt.notify_segment_ready(("layer", 0))
            opt.step()
            assert not opt._pending and not opt._armed
            opt.zero_grad()
            assert all(p.grad is None for p in params)
        flat = torch.cat([p.data.reshape(-1) for p in params])
        torch.save(flat, os.path.join(tmp, f"rank{rank}.pt"))
        dist.barrier()                                         # no rank tears its sockets down while a peer is still inside a collective
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,dtype,overlap,tensor_coll", [
    (2, torch.float32, False, False), (2, torch.bfloat16, True, False), (3, torch.float32, True, False), (4, torch.bfloat16, True, False),
    (8, torch.float32, True, False), (8, torch.bfloat16, False, False),
    # the branch RCCL runs on a node: in-place tensor collectives on slices of the flat gradient / parameter buffers
    (2, torch.float32, True, True), (3, torch.float32, False, True), (4, torch.float32, True, True), (8, torch.float32, True, True)])
def test_n_ranks_equal_one_rank(tmp_path, world, dtype, overlap, tensor_coll):
    """overlap=True: each "layer" segment is reduced asynchronously as soon as it is announced (the RCCL reduce-scatter /
    backward overlap of the product, here gloo all-reduce), the rest at step(); the result must not depend on it.
    world 2 / 4 / 8 are the sizes of the one-node scaling run (`bench.py --gpus N`), 3 a size that divides nothing."""
    steps = 3
    mp.spawn(_worker, args=(world, _free_port(), dtype, steps, str(tmp_path), overlap, tensor_coll), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    for r in range(1, world):
        assert torch.equal(r0, torch.load(os.path.join(tmp_path, f"rank{r}.pt"))), "ranks must hold identical parameters after the all-gather"

    # single-rank reference with the same per-step gradients (step 2: last parameter has no gradient)
    params = _make_params(dtype)
    flat = torch.cat([p.data.reshape(-1) for p in params]).float()
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    n_last = params[-1].numel()
    for s in range(1, steps + 1):
        per_rank = []
        for r in range(world):
            g = torch.cat([x.reshape(-1) for x in _grads_for(r, s, params)]).to(dtype).float()
            if s == 2:
                g[-n_last:] = 0
            per_rank.append(g)
        summed = per_rank[0]
        for g in per_rank[1:]:                                       # gloo sums in rank order; bf16 rounds after every addition
            summed = summed + g
            if dtype == torch.bfloat16:
                summed = summed.to(dtype).float()
        coef = min(1.0, 1.0 / (float(summed.norm()) / world + 1e-6)) / world
        R.adamw_step(flat, summed, m, v, s, 1e-2, 0.9, 0.95, 1e-8, 0.1, grad_scale=coef)
    want = flat.to(dtype)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1e-2)
    torch.testing.assert_close(r0.float(), want.float(), **tol)


def test_single_process_world1(tmp_path):
    from metamorph_amd.zero2 import Zero2AdamW
    params = _make_params(torch.float32)
    before = [p.detach().clone() for p in params]
    opt = Zero2AdamW(params, lr=1e-2, shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip)
    assert all(torch.equal(p.data, b) for p, b in zip(params, before)), "flattening must not change values"
    for p, g in zip(params, _grads_for(0, 1, params)):
        p._mm_grad_buf.copy_(g)
        p.grad = p._mm_grad_buf
    opt.step()
    assert not any(torch.equal(p.data, b) for p, b in zip(params, before))
    sd = opt.state_dict()
    opt2 = Zero2AdamW(_make_params(torch.float32), lr=1e-2, shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.master, opt.master) and opt2._step == 1
    assert all(torch.equal(a.data, b.data) for a, b in zip(opt2.params, opt.params))       # parameters follow the restored master copy


# ------------------------------------------------------------------ row N3: world-size independent optimizer checkpoints
class _Holder(torch.nn.Module):
    def __init__(self, params):
        super().__init__()
        self.ps = torch.nn.ParameterList(params)


def _ckpt_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metamorph_amd.checkpoint import consolidate_optimizer_state
        from metamorph_amd.zero2 import Zero2AdamW
        params = _make_params(torch.float32)
        for p, key in zip(params, _SEGMENTS):
            if key is not None:
                p._mm_segment = key
        model = _Holder(params)
        opt = Zero2AdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1, shard_update=_oracle_update, sumsq=_oracle_sumsq,
                         clip_coef=_oracle_clip)
        for s in (1, 2):
            for p, g in zip(params, _grads_for(rank, s, params)):
                p._mm_grad_buf.copy_(g); p.grad = p._mm_grad_buf
            opt.step(); opt.zero_grad()
        ck = consolidate_optimizer_state(opt, model)
        if rank == 0:
            torch.save(ck, os.path.join(tmp, "opt.pt"))
            torch.save(torch.cat([p.data.reshape(-1) for p in params]), os.path.join(tmp, "params2.pt"))
        else:
            assert ck is None
        for p, g in zip(params, _grads_for(rank, 3, params)):
            p._mm_grad_buf.copy_(g); p.grad = p._mm_grad_buf
        opt.step()
        if rank == 0:
            torch.save(torch.cat([p.data.reshape(-1) for p in params]), os.path.join(tmp, "params3.pt"))
        dist.barrier()                                         # no rank tears its sockets down while a peer is still inside a collective
    finally:
        dist.destroy_process_group()


def test_consolidated_optimizer_state_resumes_at_other_world_size(tmp_path):
    """Two ranks (three segments) train two steps and write a consolidated optimizer checkpoint; ONE process without segments
    loads it, applies step 3 on the summed gradient of both ranks and must land where the two ranks land."""
    from metamorph_amd.checkpoint import load_consolidated_optimizer_state
    from metamorph_amd.zero2 import Zero2AdamW
    world = 2
    mp.spawn(_ckpt_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ck = torch.load(os.path.join(tmp_path, "opt.pt"))
    assert ck["step"] == 2 and set(ck["state"]) == {f"ps.{i}" for i in range(6)}
    assert ck["state"]["ps.0"]["exp_avg"].shape == (8, 16) and ck["state"]["ps.0"]["master"].dtype == torch.float32
    params = _make_params(torch.float32)
    model = _Holder(params)

    def upd(p32, m, v, g, p_out, lr, b1, b2, eps, wd, step, scale_dev):      # two ranks' mean: the sum arrives, scale 1/2 folded in
        _oracle_update(p32, m, v, g, p_out, lr, b1, b2, eps, wd, step, scale_dev)

    opt = Zero2AdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1, shard_update=upd, sumsq=_oracle_sumsq, clip_coef=_oracle_clip)
    load_consolidated_optimizer_state(opt, model, ck)
    assert torch.allclose(torch.cat([p.data.reshape(-1) for p in params]), torch.load(os.path.join(tmp_path, "params2.pt")))
    # step 3 with the MEAN gradient of the two ranks (world 1 divides by 1)
    g0, g1 = _grads_for(0, 3, params), _grads_for(1, 3, params)
    for p, a, b in zip(params, g0, g1):
        p._mm_grad_buf.copy_((a + b) / 2); p.grad = p._mm_grad_buf
    opt.step()
    got = torch.cat([p.data.reshape(-1) for p in params])
    torch.testing.assert_close(got, torch.load(os.path.join(tmp_path, "params3.pt")), rtol=1e-5, atol=1e-6)


def _resume_worker(rank, world, port, tmp, w1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metamorph_amd.checkpoint import load_consolidated_optimizer_state
        from metamorph_amd.zero2 import Zero2AdamW
        params = _make_params(torch.float32)
        for p, key in zip(params, _SEGMENTS):
            if key is not None:
                p._mm_segment = key
        with torch.no_grad():
            for p in params:                                     # other starting weights: everything must come from the checkpoint
                p.add_(0.25)
        model = _Holder(params)
        opt = Zero2AdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1, shard_update=_oracle_update, sumsq=_oracle_sumsq,
                         clip_coef=_oracle_clip)
        load_consolidated_optimizer_state(opt, model, torch.load(os.path.join(tmp, "opt.pt")))
        assert opt._step == 2
        torch.testing.assert_close(torch.cat([p.data.reshape(-1) for p in params]), torch.load(os.path.join(tmp, "params2.pt")), rtol=0, atol=0)
        # step 3: every rank of the NEW world brings the mean gradient of the OLD world's ranks, so the mean over ranks is that mean again
        mean = [sum(gs) / w1 for gs in zip(*[_grads_for(r, 3, params) for r in range(w1)])]
        for p, g in zip(params, mean):
            p._mm_grad_buf.copy_(g); p.grad = p._mm_grad_buf
        opt.step()
        torch.save(torch.cat([p.data.reshape(-1) for p in params]), os.path.join(tmp, f"resumed_rank{rank}.pt"))
        dist.barrier()                                         # no rank tears its sockets down while a peer is still inside a collective
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w1,w2", [(2, 4), (8, 2), (4, 8)])
def test_consolidated_optimizer_state_moves_between_world_sizes(tmp_path, w1, w2):
    """w1 ranks train two steps and consolidate; w2 ranks (other shard boundaries, other starting weights) load the checkpoint, take step 3
    and must hold what the w1 ranks hold after their step 3 -- the role DeepSpeed's zero_to_fp32 + a fresh launch play for the reference."""
    mp.spawn(_ckpt_worker, args=(w1, _free_port(), str(tmp_path)), nprocs=w1, join=True)
    mp.spawn(_resume_worker, args=(w2, _free_port(), str(tmp_path), w1), nprocs=w2, join=True)
    want = torch.load(os.path.join(tmp_path, "params3.pt"))
    for r in range(w2):
        torch.testing.assert_close(torch.load(os.path.join(tmp_path, f"resumed_rank{r}.pt")), want, rtol=2e-5, atol=2e-6)


# ------------------------------------------------------------------ the real module tree: segments, hook wiring, overlap
def _model_worker(rank, world, port, tmp, overlap, tensor_coll=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if tensor_coll:
        __import__("metamorph_amd.zero2", fromlist=["x"]).set_collective_mode(tensor_collectives=True)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metamorph_amd import functional as F
        from metamorph_amd.factory import build_model
        from metamorph_amd.zero2 import Zero2AdamW, tag_segments
        torch.manual_seed(0)
        llm = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=1,
                   vocab_size=300, rms_norm_eps=1e-5, rope_theta=500000.0)
        model = build_model(llm, dict(num_hidden_layers=1, intermediate_size=144, image_size=28), num_image_tokens=4)
        params = [p for p in model.parameters() if p.requires_grad]
        tag_segments(model)
        opt = Zero2AdamW(params, lr=1e-2, shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip,
                         overlap=overlap).enable_overlap()
        layers = model.get_model().layers
        # embeddings | layer 0 | layer 1 | layer 2 | final norm, projector, lm_head, vision head
        assert len(opt.segs) == 5 and [sg["key"] for sg in opt.segs[1:4]] == [("layer", i) for i in range(3)]
        for i, layer in enumerate(layers):                   # a layer's parameters are exactly its segment, fused blocks adjacent
            assert {id(p) for p in layer.parameters()} == {id(p) for p in opt.segs[1 + i]["params"]}
            att = layer.self_attn
            assert att.k_proj.weight.data_ptr() == att.q_proj.weight.data_ptr() + att.q_proj.weight.numel() * 2
        for step in (1, 2):
            opt.arm_overlap()
            g = torch.Generator().manual_seed(100 * step + rank)
            # the backward pass: head / norm first, decoder layers last to first (each announces itself), embeddings last
            order = [p for p in params if getattr(p, "_mm_segment", None) is None]
            for p in order:
                p._mm_grad_buf.copy_(torch.randn(p.shape, generator=g).to(p.dtype)); p.grad = p._mm_grad_buf
            for i in reversed(range(len(layers))):
                for p in layers[i].parameters():
                    p._mm_grad_buf.copy_(torch.randn(p.shape, generator=g).to(p.dtype)); p.grad = p._mm_grad_buf
                F._LAYER_GRAD_HOOK(layers[i])                # what DecoderLayerFn.backward does at its end
                assert ((1 + i) in opt._pending) == bool(overlap)
            opt.step()
            opt.zero_grad()
        torch.save(opt.flat_param.clone(), os.path.join(tmp, f"model_rank{rank}_ov{int(overlap)}.pt"))
        F.set_layer_grad_hook(None)
        dist.barrier()                                         # no rank tears its sockets down while a peer is still inside a collective
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tensor_coll", [(2, False), (8, True)])
def test_overlap_on_the_real_module_tree(tmp_path, world, tensor_coll):
    """tag_segments + enable_overlap on the actual MetaMorph module tree (CPU parameters, gloo): one segment per decoder
    layer, announcements start that layer's reduction, and the result is bit-identical to the non-overlapped schedule."""
    for ov in (False, True):
        mp.spawn(_model_worker, args=(world, _free_port(), str(tmp_path), ov, tensor_coll), nprocs=world, join=True)
    a = [torch.load(os.path.join(tmp_path, f"model_rank{r}_ov0.pt")) for r in range(world)]
    b = [torch.load(os.path.join(tmp_path, f"model_rank{r}_ov1.pt")) for r in range(world)]
    assert all(torch.equal(a[0], x) for x in a[1:] + b)


def test_param_groups_with_their_own_learning_rate():
    """torch-style parameter groups (the reference's `vision_lr` group, metamorph_trainer.py:201-233): each group is updated with
    its own lr / weight decay, a segment never spans two groups, and the result equals per-group oracle AdamW steps under one
    global clipping coefficient."""
    from metamorph_amd.zero2 import Zero2AdamW
    params = _make_params(torch.float32)
    ref = [p.detach().clone() for p in params]
    groups = [dict(params=params[:4], lr=1e-2, weight_decay=0.1), dict(params=params[4:], lr=1e-3, weight_decay=0.0)]
    opt = Zero2AdamW(groups, betas=(0.9, 0.95), shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip)
    assert len(opt.segs) == 2 and [sg["group"] for sg in opt.segs] == [0, 1]
    m = [torch.zeros_like(p) for p in ref]
    v = [torch.zeros_like(p) for p in ref]
    for step in (1, 2):
        grads = _grads_for(0, step, params)
        for p, g in zip(params, grads):
            p._mm_grad_buf.copy_(g); p.grad = p._mm_grad_buf
        opt.step(); opt.zero_grad()
        norm = float(torch.sqrt(sum((g.float() ** 2).sum() for g in grads)))
        coef = min(1.0, 1.0 / (norm + 1e-6))
        for i, (r, g) in enumerate(zip(ref, grads)):
            lr, wd = (1e-2, 0.1) if i < 4 else (1e-3, 0.0)
            R.adamw_step(r, g, m[i], v[i], step, lr, 0.9, 0.95, 1e-8, wd, grad_scale=coef)
    for p, r in zip(params, ref):
        torch.testing.assert_close(p.data, r, rtol=1e-5, atol=1e-6)


def _group_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metamorph_amd.zero2 import Zero2AdamW
        params = _make_params(torch.float32)
        for p, key in zip(params, _SEGMENTS):
            if key is not None:
                p._mm_segment = key
        # the group boundary cuts THROUGH "layer 0" (tensors 0,1 | 2): that key then owns two segments
        groups = [dict(params=params[:2], lr=1e-2, weight_decay=0.1), dict(params=params[2:], lr=1e-3, weight_decay=0.0)]
        opt = Zero2AdamW(groups, betas=(0.9, 0.95), shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip,
                         overlap=True)
        assert [sg["group"] for sg in opt.segs] == [0, 1, 1, 1] and opt.seg_of_key[("layer", 0)] == [0, 1]
        for s in (1, 2):
            opt.arm_overlap()
            grads = _grads_for(rank, s, params)
            for idx in (5, 4, 3, 2, 1, 0):
                params[idx]._mm_grad_buf.copy_(grads[idx])
                params[idx].grad = params[idx]._mm_grad_buf
                if idx == 3:
                    opt.notify_segment_ready(("layer", 1))
                if idx == 0:
                    opt.notify_segment_ready(("layer", 0))
                    assert {0, 1} <= set(opt._pending)            # both halves of the cut layer went out
            opt.step()
            opt.zero_grad()
        torch.save(torch.cat([p.data.reshape(-1) for p in params]), os.path.join(tmp, f"grp_rank{rank}.pt"))
        dist.barrier()                                         # no rank tears its sockets down while a peer is still inside a collective
    finally:
        dist.destroy_process_group()


def test_param_groups_on_two_ranks(tmp_path):
    """Two ranks, two parameter groups whose boundary cuts through a tagged layer: every rank ends with the same parameters,
    equal to per-group oracle AdamW steps on the MEAN gradient under one global clipping coefficient."""
    world = 2
    mp.spawn(_group_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"grp_rank{r}.pt")) for r in range(world)]
    assert torch.equal(got[0], got[1])
    params = _make_params(torch.float32)
    ref = [p.detach().clone() for p in params]
    m = [torch.zeros_like(p) for p in ref]
    v = [torch.zeros_like(p) for p in ref]
    for step in (1, 2):
        per_rank = [_grads_for(r, step, params) for r in range(world)]
        mean = [sum(gs) / world for gs in zip(*per_rank)]
        norm = float(torch.sqrt(sum((g.float() ** 2).sum() for g in mean)))
        coef = min(1.0, 1.0 / (norm + 1e-6))
        for i, (r, g) in enumerate(zip(ref, mean)):
            lr, wd = (1e-2, 0.1) if i < 2 else (1e-3, 0.0)
            R.adamw_step(r, g, m[i], v[i], step, lr, 0.9, 0.95, 1e-8, wd, grad_scale=coef)
    torch.testing.assert_close(got[0], torch.cat([r.reshape(-1) for r in ref]), rtol=1e-5, atol=1e-6)
