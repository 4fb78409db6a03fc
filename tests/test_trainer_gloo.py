"""`MetaMorphTrainer` plumbing on CPU, world_size 2 over gloo: HF `Trainer` must NOT wrap the model in
DistributedDataParallel (the decoder's autograd nodes write weight gradients straight into Zero2AdamW's flat buffer, a DDP reducer
would wait for hooks that never fire), `arm_overlap()` must be called on the last micro-step of every accumulation window, clipping
must happen inside the sharded optimizer, and two ranks must end up with the parameters one rank gets on the union of their batches.
The model is a stand-in (the HIP kernels cannot run here); the shard arithmetic is injected from the oracle like in test_zero2_gloo."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from test_zero2_gloo import _free_port, _oracle_clip, _oracle_sumsq, _oracle_update


class Toy(nn.Module):
    """Two 'decoder layers' + a head; `get_model()` marks it as a model whose gradients Zero2AdamW exchanges."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.layers = nn.ModuleList([nn.Linear(8, 8), nn.Linear(8, 8)])
        self.head = nn.Linear(8, 1)

    def get_model(self):
        return self

    def forward(self, x=None, y=None):
        h = x
        for l in self.layers:
            h = torch.tanh(l(h))
        return {"loss": ((self.head(h)[:, 0] - y) ** 2).mean()}


class _DS(torch.utils.data.Dataset):
    def __init__(self, n):
        g = torch.Generator().manual_seed(5)
        self.x, self.y = torch.randn(n, 8, generator=g), torch.randn(n, generator=g)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return {"x": self.x[i], "y": self.y[i]}


def _train(out_dir, per_device_bs, accum, steps):
    from transformers import TrainingArguments
    from metamorph_amd.trainer import MetaMorphTrainer
    from metamorph_amd.zero2 import Zero2AdamW
    armed = []

    class SeqTrainer(MetaMorphTrainer):
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    orig = Zero2AdamW.arm_overlap
    Zero2AdamW.arm_overlap = lambda self: (armed.append(1), orig(self))[1]
    try:
        args = TrainingArguments(output_dir=out_dir, per_device_train_batch_size=per_device_bs, gradient_accumulation_steps=accum,
                                 max_steps=steps, learning_rate=1e-2, weight_decay=0.1, max_grad_norm=0.5, lr_scheduler_type="constant",
                                 use_cpu=True, report_to=[], save_strategy="no", logging_steps=1, remove_unused_columns=False,
                                 dataloader_num_workers=0, dataloader_pin_memory=False, seed=3)
        model = Toy()
        tr = SeqTrainer(model=model, args=args, train_dataset=_DS(64), zero2_kwargs=dict(
            shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip))
        tr.train()
    finally:
        Zero2AdamW.arm_overlap = orig
    z = tr._zero2()
    assert isinstance(z, Zero2AdamW) and z._step == steps and z.max_grad_norm == 0.5 and tr.args.max_grad_norm == 0.0
    # optimizer checkpoint through the Trainer's own hooks: the WHOLE state (every rank's shard), restorable into a fresh optimizer
    ck = os.path.join(os.path.dirname(out_dir), "opt_ckpt_" + os.path.basename(out_dir).lstrip("r0123456789"))
    tr._save_optimizer_and_scheduler(ck)
    if dist.is_initialized():
        dist.barrier()
    tr2 = SeqTrainer(model=Toy(), args=args, train_dataset=_DS(64), zero2_kwargs=dict(
        shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip))
    tr2.create_optimizer_and_scheduler(num_training_steps=steps)
    tr2._load_optimizer_and_scheduler(ck)
    z2 = tr2._zero2()
    assert z2._step == z._step
    for k in ("master", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(z2, k), getattr(z, k)), k
    for pa, pb in zip(tr2.model.parameters(), model.parameters()):       # the bf16/fp32 parameters follow the restored master copy
        assert torch.equal(pa.data, pb.data)
    saved = torch.load(os.path.join(ck, "optimizer.pt"), weights_only=True)
    assert set(saved["state"]) == {n for n, _ in model.named_parameters()}  # per-name, world-size independent
    assert type(tr.model_wrapped) is Toy and type(tr.model) is Toy, type(tr.model_wrapped)        # no DDP wrapper
    assert len(armed) == steps, (len(armed), steps)                                                # once per accumulation window
    assert len(z.segs) >= 3                                                                        # layers tagged as segments
    # weight decay groups: biases in the no-decay group
    wd = {id(p): g["weight_decay"] for g in z.param_groups for p in g["params"]}
    assert wd[id(model.head.bias)] == 0.0 and wd[id(model.head.weight)] == 0.1
    return torch.cat([p.data.reshape(-1) for p in model.parameters()])


def test_trainer_dataloader_keeps_batches_on_the_host_until_prepare_inputs(tmp_path):
    """accelerate's prepared DataLoader must NOT place batches on the device itself (device_placement=False), or `_prepare_inputs`
    never sees a host original to register as a mirror and every step pays the `.cpu()` synchronisation.  Checked on the prepared loader
    HF's own `get_train_dataloader` returns; the device half (STATS['sync'] stays 0 over a Trainer run) is tests/test_trainer_gpu.py."""
    from transformers import TrainingArguments
    from metamorph_amd.trainer import MetaMorphTrainer
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, max_steps=1, use_cpu=True, report_to=[], save_strategy="no",
                             remove_unused_columns=False, dataloader_num_workers=0, dataloader_pin_memory=False)
    tr = MetaMorphTrainer(model=Toy(), args=args, train_dataset=_DS(8), zero2_kwargs=dict(
        shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip))
    dl = tr.get_train_dataloader()
    assert hasattr(dl, "device") and dl.device is None, getattr(dl, "device", "no attribute")     # accelerate: device None <=> no placement
    batch = next(iter(dl))
    assert all(v.device.type == "cpu" for v in batch.values())
    from accelerate import Accelerator
    plain = Accelerator(cpu=True).prepare(torch.utils.data.DataLoader(_DS(8), batch_size=2))
    assert plain.device is not None                               # what the loader looks like without the override (placement on)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      ACCELERATE_USE_CPU="true")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat = _train(os.path.join(tmp, f"r{rank}"), per_device_bs=2, accum=2, steps=3)
        torch.save(flat, os.path.join(tmp, f"rank{rank}.pt"))
        dist.barrier()                                         # no rank tears its sockets down while a peer is still inside a collective
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_trainer_n_ranks_no_ddp_equals_one_rank(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(tmp_path / "rank0.pt")
    for r in range(1, world):
        assert torch.equal(a, torch.load(tmp_path / f"rank{r}.pt"))
    # one rank, micro-batches of 2 * world = the union of what the ranks saw per micro-step (accelerate hands batch world * k + r to
    # rank r); mean loss over equal-sized parts == mean over the union
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    os.environ["ACCELERATE_USE_CPU"] = "true"
    try:
        one = _train(str(tmp_path / "one"), per_device_bs=2 * world, accum=2, steps=3)
    finally:
        os.environ.pop("ACCELERATE_USE_CPU", None)
    torch.testing.assert_close(a, one, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ ZeRO-3: model save is a collective
def _accum(dst, src, first):
    if first:
        dst.copy_(src)
    else:
        dst.add_(src)


def _zero3_save_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      ACCELERATE_USE_CPU="true")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from transformers import TrainingArguments
        from metamorph_amd.trainer import MetaMorphTrainer
        from metamorph_amd.zero3 import Zero3AdamW
        out = os.path.join(tmp, "run")
        args = TrainingArguments(output_dir=out, per_device_train_batch_size=2, max_steps=3, learning_rate=1e-2, use_cpu=True, report_to=[],
                                 save_strategy="no", remove_unused_columns=False, dataloader_pin_memory=False, seed=3)
        model = Toy()
        want = {k: v.clone() for k, v in model.state_dict().items()}
        tr = MetaMorphTrainer(model=model, args=args, train_dataset=_DS(16), zero_stage=3, zero2_kwargs=dict(
            shard_update=_oracle_update, sumsq=_oracle_sumsq, clip_coef=_oracle_clip, accumulate=_accum, min_shard_numel=1, param_slots=2,
            grad_slots=1))
        tr.create_optimizer_and_scheduler(num_training_steps=3)
        z = tr._zero2()
        assert isinstance(z, Zero3AdamW) and z.world == 2
        assert all(p.data.numel() == 0 for l in model.layers for p in l.parameters())        # sharded: the tree holds no layer weights
        # (1) the user-facing save: HF calls _save on the should_save rank only -- the gather behind it must be entered by EVERY rank
        tr.save_model(os.path.join(tmp, "final"))
        dist.barrier()
        # (2) a save_steps checkpoint (model + optimizer shards + scheduler + trainer state) goes through the same path
        tr.state.global_step = 2
        tr._save_checkpoint(model, trial=None)
        dist.barrier()
        # (3) the next collective pairs up correctly (a stray all_gather from a one-rank save would be consumed here)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        assert float(t) == 3.0
        if rank == 0:
            import glob
            for d in (os.path.join(tmp, "final"), os.path.join(out, "checkpoint-2")):
                files = glob.glob(os.path.join(d, "*.safetensors")) + glob.glob(os.path.join(d, "pytorch_model*.bin"))
                assert files, os.listdir(d)
                if files[0].endswith(".bin"):
                    got = torch.load(files[0], weights_only=True)
                else:
                    from safetensors.torch import load_file
                    got = load_file(files[0])
                assert set(got) == set(want)
                for k in want:
                    assert torch.equal(got[k], want[k]), k                                   # the complete tensors, not one rank's slice
            ck = os.path.join(out, "checkpoint-2")
            assert os.path.isfile(os.path.join(ck, "zero3_rank0-of-2-optimizer.pt")) and os.path.isfile(os.path.join(ck, "zero3_rank1-of-2-optimizer.pt"))
        # the direct _save without a gathered state dict refuses instead of hanging the other ranks
        with pytest.raises(RuntimeError):
            tr._save(os.path.join(tmp, "bad"))
        torch.save(torch.ones(1), os.path.join(tmp, f"ok{rank}.pt"))
        dist.barrier()                                         # no rank tears its sockets down while a peer is still inside a collective
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_zero3_trainer_save_model_is_collective(tmp_path):
    port = _free_port()
    mp.spawn(_zero3_save_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0.pt").exists() and (tmp_path / "ok1.pt").exists()
