"""HF `Trainer` drives the HIP model the way the reference's launch scripts do (boundary row (b)): `--gradient_checkpointing True`
(scripts/*.sh), gradient accumulation, `MetaMorphTrainer.create_optimizer` -> sharded AdamW.  Needs an MI355X:  pytest -m gpu"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402
from oracle.ref_model import init_state_dict  # noqa: E402
from test_model_gpu import DEV, T, hip_model, tiny_cfg  # noqa: E402


class _Rows(torch.utils.data.Dataset):
    def __init__(self, g, reps):
        ids, lab, msk = T(g["input_ids"]), T(g["labels"]), T(g["attention_mask"])
        imgs = T(g["images"])
        self.items = []
        k = 0
        for b in range(ids.shape[0]):
            n = int(msk[b].sum())
            n_img = max(1, int((ids[b, :n] == -200).sum()))          # a text-only sample carries its dummy image (train.py:1230-1240)
            self.items.append(dict(input_ids=ids[b, :n].clone(), labels=lab[b, :n].clone(), image=[imgs[k + j].bfloat16() for j in range(n_img)]))
            k += n_img
        self.items = self.items * reps

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _collator():
    from metamorph_amd.data import DataCollatorForSupervisedDataset
    from oracle.fake_tokenizer import FakeTokenizer
    return DataCollatorForSupervisedDataset(tokenizer=FakeTokenizer(model_max_length=64))


@pytest.mark.parametrize("zero_stage", [2, 3])
def test_hf_trainer_two_steps_with_gradient_checkpointing(tmp_path, zero_stage):
    from transformers import TrainingArguments
    from metamorph_amd.trainer import MetaMorphTrainer
    from metamorph_amd.zero2 import Zero2AdamW, tag_segments
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    ds = _Rows(g, reps=2)                                            # 6 samples: 3 micro-batches of 2, accumulation 3 -> 1 optimizer step / epoch
    collate = _collator()

    class SeqTrainer(MetaMorphTrainer):                              # fixed sample order so the hand-written loop below sees the same batches
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, gradient_accumulation_steps=3, max_steps=2,
                             learning_rate=1e-3, weight_decay=0.01, max_grad_norm=1.0, lr_scheduler_type="constant", bf16=True,
                             gradient_checkpointing=True, report_to=[], save_strategy="no", logging_steps=1,
                             remove_unused_columns=False, dataloader_num_workers=0, dataloader_pin_memory=False)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    # zero_stage=3: the reference's scripts/zero3.json recipe -- decoder-layer parameters sharded, gathered per layer through the hooks
    trainer = SeqTrainer(model=model, args=args, train_dataset=ds, data_collator=collate, zero_stage=zero_stage,
                         zero2_kwargs=dict(min_shard_numel=1) if zero_stage == 3 else None)
    from metamorph_amd import hostmirror
    before = dict(hostmirror.STATS)
    out = trainer.train()
    # the Trainer's own dataloader hands HOST batches to _prepare_inputs, which registers them as mirrors of the device tensors:
    # six micro-batches built their splice plans without one device -> host copy
    assert hostmirror.STATS["sync"] == before["sync"] and hostmirror.STATS["mirror"] >= before["mirror"] + 3 * 6, (before, hostmirror.STATS)
    z = trainer._zero2()
    from metamorph_amd.zero3 import Zero3AdamW
    assert isinstance(z, Zero3AdamW if zero_stage == 3 else Zero2AdamW) and z._step == 2
    from metamorph_amd import functional as F
    F.set_layer_grad_hook(None)
    F.set_param_ready_hook(None)
    # optimizer checkpoint through the Trainer's hooks: the whole sharded state (ZeRO-2: one per-name, world-size independent file;
    # ZeRO-3: one shard file per rank), restored into a fresh optimizer over DIFFERENT initial weights
    ck = str(tmp_path / "opt_ckpt")
    trainer._save_optimizer_and_scheduler(ck)
    m2 = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]) + 1, dtype=torch.bfloat16))
    t2 = SeqTrainer(model=m2, args=args, train_dataset=ds, data_collator=collate, zero_stage=zero_stage,
                    zero2_kwargs=dict(min_shard_numel=1) if zero_stage == 3 else None)
    t2.create_optimizer_and_scheduler(num_training_steps=2)
    t2._load_optimizer_and_scheduler(ck)
    z2 = t2._zero2()
    assert z2 is not z and z2._step == 2
    for k in ("master", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(z2, k), getattr(z, k)), k
    if zero_stage == 2:                                     # the bf16 parameters follow the restored master copy
        for (n, pa), (_, pb) in zip(m2.named_parameters(), model.named_parameters()):
            if pa.requires_grad:
                assert torch.equal(pa.data, pb.data), n
    else:
        for sa, sb in zip(z2.segs, z.segs):
            assert torch.equal(sa["p_shard"] if sa["sharded"] else sa["my_param"], sb["p_shard"] if sb["sharded"] else sb["my_param"])
    del m2, t2, z2
    F.set_layer_grad_hook(None)
    F.set_param_ready_hook(None)
    if zero_stage == 3:                                     # compare the gathered full parameters below
        full = z.gather_full_parameters()
        for p_, t_ in full.items():
            p_.data = t_
    assert model.model.gradient_checkpointing and model.is_gradient_checkpointing    # HF's gradient_checkpointing_enable() was honoured
    assert type(trainer.model_wrapped) is type(model)                                 # no DDP / DataParallel wrapper
    assert np.isfinite(out.training_loss)
    # the same two optimizer steps by hand (no checkpointing): recompute runs the same kernels on the same inputs => identical parameters
    ref = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    ref.train()
    tag_segments(ref)
    from metamorph_amd.trainer import optimizer_grouped_parameters
    opt = Zero2AdamW(optimizer_grouped_parameters(ref, 0.01), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0)
    for _ in range(2):
        opt.zero_grad()
        for mb in range(3):
            batch = collate([ds[2 * mb], ds[2 * mb + 1]])
            o = ref(input_ids=batch["input_ids"].to(DEV), attention_mask=batch["attention_mask"].to(DEV), labels=batch["labels"].to(DEV),
                    images=batch["images"].to(DEV))
            (o.loss / 3).backward()
        opt.step()
    torch.cuda.synchronize()
    # Identical kernels on identical inputs, and no kernel on the update path uses atomics (the grad-norm partials and the bias-gradient
    # column sums are reduced in a fixed order): under ZeRO-2 the Trainer-driven, checkpointed run and the hand-written loop end with
    # bit-identical parameters and optimizer state.
    bad = []
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if p.requires_grad:
            a, b = p.data.float(), q.data.float()
            same = 1.0 if torch.equal(a, b) else min(float((a == b).double().mean()), 1.0 - 1e-12)
            rel_l2 = float((a - b).norm() / b.norm().clamp_min(1e-20))
            # (ZeRO-3 folds every micro-step's reduce-scattered bf16 slice into the gradient shard: one more bf16 rounding per micro-step
            # than ZeRO-2's in-epilogue accumulation, so with accumulation 3 its parameters agree to bf16 noise, not bit for bit)
            lim_same, lim_rel = (1.0, 0.0) if zero_stage == 2 else (0.85, 3e-3)
            if same < lim_same or rel_l2 > lim_rel:
                bad.append((n, same, rel_l2))
    assert not bad, f"Trainer-driven (checkpointed) and hand-written steps differ (name, fraction bit-equal, rel L2): {bad[:8]} ({len(bad)} tensors)"
    # first moments after two identical steps: m = 0.1 * (0.9 g1 + g2) * coef -- equal to bf16-noise level over the whole shard
    zr = opt
    if zero_stage == 2:
        assert torch.equal(z.exp_avg, zr.exp_avg) and torch.equal(z.exp_avg_sq, zr.exp_avg_sq)
    from metamorph_amd import functional as F
    F.set_layer_grad_hook(None)
    F.set_param_ready_hook(None)


def test_gradient_checkpointing_recompute_is_bit_identical():
    g = np.load(os.path.join(GOLDEN, "e2e_multi_frame_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    batch = dict(input_ids=T(g["input_ids"]).to(DEV), attention_mask=T(g["attention_mask"]).to(DEV), labels=T(g["labels"]).to(DEV),
                 images=T(g["images"]).to(DEV).bfloat16())
    grads = []
    for ckpt in (False, True, 1):                                  # no recompute / every layer (the reference's mode) / the first decoder layer only
        model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
        model.train()
        tower = model.get_model().vision_tower                     # the trainable-tower layers recompute as well
        tower.freeze_vision = False
        for n, p in tower.named_parameters():
            p.requires_grad_("post_layernorm" not in n)
        if ckpt:
            model.gradient_checkpointing_enable()
            if ckpt is not True:                                   # MI355X extension: only the first n decoder layers keep just their input
                model.get_model().checkpoint_layers = int(ckpt)
        torch.cuda.reset_peak_memory_stats()
        out = model(**batch)
        out.loss.backward()
        grads.append((float(out.loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * abs(grads[0][0])      # the scalar loss is an fp32 atomicAdd over rows: equal to rounding
    assert grads[0][1].keys() == grads[1][1].keys() and len(grads[0][1]) > 40
    for n in grads[0][1]:
        assert torch.equal(grads[0][1][n], grads[1][1][n]), n
        assert torch.equal(grads[0][1][n], grads[2][1][n]), n
    # eval / no_grad never recomputes and a disabled flag restores the saved-activation path
    model.gradient_checkpointing_disable()
    assert not model.model.gradient_checkpointing
