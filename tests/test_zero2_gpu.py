"""Zero2AdamW on the GPU with the HIP shard-update / norm kernels (world size 1) against the oracle AdamW, plus a
full optimizer step of the tiny model.  Needs an MI355X:  pytest -m gpu"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402


def test_zero2_step_matches_oracle_adamw():
    from metamorph_amd.zero2 import Zero2AdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 32), (16, 32), (16, 32), (32, 64), (777,)]
    params = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).bfloat16().cuda()) for s in shapes]
    ref = torch.cat([p.detach().float().cpu().reshape(-1) for p in params])
    m, v = torch.zeros_like(ref), torch.zeros_like(ref)
    opt = Zero2AdamW(params, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, max_grad_norm=1.0)
    assert params[1].data_ptr() == params[0].data_ptr() + params[0].numel() * 2       # flat, unpadded layout
    for step in range(1, 4):
        grads = [(torch.randn(*s, generator=g) * 2).bfloat16() for s in shapes]
        for p, gr in zip(params, grads):
            p._mm_grad_buf.copy_(gr.cuda())
            p.grad = p._mm_grad_buf
        opt.step()
        flat_g = torch.cat([x.float().reshape(-1) for x in grads])
        coef = min(1.0, 1.0 / (float(flat_g.norm()) + 1e-6))
        R.adamw_step(ref, flat_g, m, v, step, 1e-2, 0.9, 0.95, 1e-8, 0.1, grad_scale=coef)
        assert abs(opt.grad_norm_value() - float(flat_g.norm())) < 1e-3 * float(flat_g.norm())
        opt.zero_grad()
    got = torch.cat([p.detach().float().cpu().reshape(-1) for p in params])
    torch.testing.assert_close(opt.master.cpu()[: ref.numel()], ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got, ref.bfloat16().float(), rtol=0, atol=1e-2)


def test_model_training_steps_reduce_loss():
    """Three optimizer steps of the tiny model on one batch: loss goes down and parameters stay finite."""
    import os
    from conftest import GOLDEN
    from test_model_gpu import T, hip_model, tiny_cfg
    from oracle.ref_model import init_state_dict
    from metamorph_amd.zero2 import Zero2AdamW
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
    model.train()
    opt = Zero2AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-3, weight_decay=0.0, max_grad_norm=1.0)
    args = dict(input_ids=T(g["input_ids"]).cuda(), attention_mask=T(g["attention_mask"]).cuda(),
                labels=T(g["labels"]).cuda(), images=T(g["images"]).cuda().bfloat16())
    losses = []
    for _ in range(4):
        opt.zero_grad()
        out = model(**args)
        out.loss.backward()
        opt.step()
        losses.append(float(out.loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.5, losses
    assert all(torch.isfinite(p.float()).all() for p in model.parameters())


def test_tied_embeddings_train_as_one_parameter_under_zero2():
    """`tie_word_embeddings` (LLaMA-3.2 1B / 3B bases; round 6) under the sharded optimizer: lm_head.weight IS embed_tokens.weight -- one
    Parameter, one slice of the flat buffers, one gradient that receives the fused CE's dW and the splice's embedding-row sums.  Four steps
    against the oracle trained the same way (autograd through ONE shared tensor + the oracle's AdamW): loss per step to 1 %, the tie survives
    every step (both modules read the re-pointed flat storage)."""
    import os
    from conftest import GOLDEN
    from test_model_gpu import T, hip_model, tiny_cfg
    from oracle.ref_model import forward as oracle_forward, init_state_dict
    from metamorph_amd.zero2 import Zero2AdamW
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_tied_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4, tie_word_embeddings=True)
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    model = hip_model(cfg, {k: v.bfloat16() for k, v in sd.items()})
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    assert model.lm_head.weight is model.model.embed_tokens.weight and sum(p is model.lm_head.weight for p in params) == 1
    opt = Zero2AdamW(params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0)
    args = dict(input_ids=T(g["input_ids"]).cuda(), attention_mask=T(g["attention_mask"]).cuda(), labels=T(g["labels"]).cuda(),
                images=T(g["images"]).cuda().bfloat16())
    # the oracle: fp32 weights rounded through bf16 once (what the model holds), the tied tensor shared by both names
    ref = {k: v.bfloat16().float() for k, v in sd.items()}
    ref["lm_head.weight"] = ref["model.embed_tokens.weight"]
    train = {k: v for k, v in ref.items() if "vision_tower" not in k and "vision_proj" not in k and k != "lm_head.weight"}
    for v in train.values():
        v.requires_grad_(True)
    m_ = {k: torch.zeros_like(v) for k, v in train.items()}
    v_ = {k: torch.zeros_like(v) for k, v in train.items()}
    hip, ora = [], []
    for step in range(1, 5):
        opt.zero_grad()
        out = model(**args)
        out.loss.backward()
        opt.step()
        hip.append(float(out.loss.detach()))
        assert model.lm_head.weight is model.model.embed_tokens.weight
        assert model.lm_head.weight.data_ptr() == model.model.embed_tokens.weight.data_ptr()
        o = oracle_forward(ref, cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), T(g["images"]), return_logits=False)
        for v in train.values():
            v.grad = None
        o["loss"].backward()
        ora.append(float(o["loss"].detach()))
        gn = torch.sqrt(sum((v.grad.float() ** 2).sum() for v in train.values()))
        coef = min(1.0, 1.0 / (float(gn) + 1e-6))
        with torch.no_grad():
            for k, v in train.items():
                R.adamw_step(v.data.view(-1), v.grad.reshape(-1), m_[k].view(-1), v_[k].view(-1), step, 2e-3, 0.9, 0.999, 1e-8, 0.0, grad_scale=coef)
    print(f"\n   tied embeddings under Zero2AdamW: loss hip {[round(x, 4) for x in hip]} oracle {[round(x, 4) for x in ora]}")
    assert hip[-1] < hip[0] - 0.3
    for a, b in zip(hip, ora):
        assert abs(a - b) <= 1e-2 * abs(b), (hip, ora)


def test_loss_curve_matches_oracle_training():
    """north_star: "loss curves matching reference within tolerance".  Six optimizer steps of the tiny model on one mixed
    batch (text CE + answer-image cosine loss): the HIP run (bf16 activations and weights, fp32 master, Zero2AdamW with
    global-norm clipping) against the CPU oracle trained the same way in fp32 (forward / autograd of oracle.ref_model, AdamW of
    oracle.ref_ops, weights rounded to bf16 after every step like the HIP parameters).  Tolerance 1.5 % of the loss per step
    (bf16 activation noise: the reference's own bf16 and fp32 runs differ by ~1e-3 on this model at step 0)."""
    import os
    from conftest import GOLDEN
    from test_model_gpu import T, hip_model, tiny_cfg
    from oracle.ref_model import forward as oracle_forward, init_state_dict
    from metamorph_amd.zero2 import Zero2AdamW
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    seed, lr, steps = int(g["seed"]), 1e-3, 6
    batch = dict(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]))

    # ---- HIP
    model = hip_model(cfg, init_state_dict(cfg, seed=seed, dtype=torch.bfloat16))
    model.train()
    opt = Zero2AdamW([p for p in model.parameters() if p.requires_grad], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                     max_grad_norm=1.0)
    dev_batch = {k: (v.cuda().bfloat16() if k == "images" else v.cuda()) for k, v in batch.items()}
    hip_losses = []
    for _ in range(steps):
        opt.zero_grad()
        out = model(**dev_batch)
        out.loss.backward()
        opt.step()
        hip_losses.append(float(out.loss.detach()))

    # ---- oracle (fp32 master weights, bf16-rounded working weights, same freeze policy: tower frozen)
    master = init_state_dict(cfg, seed=seed)                                  # fp32
    train = [k for k in master if "vision_tower" not in k and "vision_proj" not in k]
    mom = {k: (torch.zeros_like(master[k]), torch.zeros_like(master[k])) for k in train}
    ora_losses = []
    for step in range(1, steps + 1):
        sd = {k: v.bfloat16().float() for k, v in master.items()}
        for k in train:
            sd[k].requires_grad_(True)
        o = oracle_forward(sd, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"], return_logits=False)
        o["loss"].backward()
        ora_losses.append(float(o["loss"].detach()))
        grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in train}
        norm = float(torch.sqrt(sum((gr.float() ** 2).sum() for gr in grads.values())))
        coef = min(1.0, 1.0 / (norm + 1e-6))
        for k in train:
            R.adamw_step(master[k], grads[k], mom[k][0], mom[k][1], step, lr, 0.9, 0.999, 1e-8, 0.0, grad_scale=coef)
    print("\n   loss curve  hip:", " ".join(f"{x:.4f}" for x in hip_losses), "\n            oracle:", " ".join(f"{x:.4f}" for x in ora_losses))
    assert ora_losses[-1] < ora_losses[0] - 0.3
    for s, (a, b) in enumerate(zip(hip_losses, ora_losses)):
        assert abs(a - b) <= 1.5e-2 * abs(b), f"step {s}: hip {a} vs oracle {b}"


def _curve_batches(n, seed, V, start_id):
    """`n` micro-batches of two samples each: an image-QA sample (prompt-side image) and a generation sample (answer-side image)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        L = 40
        ids = torch.randint(3, V - 10, (2, L), generator=g)
        ids[:, 0] = ids[:, 1] = 1
        ids[0, 6], ids[0, 7], ids[0, 8] = start_id, -200, start_id + 1
        lab = torch.full_like(ids, -100)
        lab[0, 20:] = ids[0, 20:]
        n1 = int(torch.randint(24, 36, (1,), generator=g))
        ids[1, n1 - 4], ids[1, n1 - 3], ids[1, n1 - 2], ids[1, n1 - 1] = start_id, -200, start_id + 1, 2
        ids[1, n1:] = 0                                              # pad id 0
        lab[1, 10:n1] = ids[1, 10:n1]
        lab[1, n1 - 3] = -200
        mask = torch.ones_like(ids, dtype=torch.bool)
        mask[1, n1:] = False
        images = torch.randn(2, 3, 56, 56, generator=g)
        out.append(dict(input_ids=ids, attention_mask=mask, labels=lab, images=images))
    return out


def test_long_loss_curve_matches_oracle_training():
    """Row LC ("loss curves matching reference within tolerance"): 50 optimizer steps of a 4-layer, h = 1024 model (8 query / 2 KV
    heads of 128, I = 2816, V = 32002) with BOTH heads live in every micro-batch, the reference's finetune recipe in miniature
    (scripts/debug_finetune_1node.sh:47-49: AdamW, weight decay 0, cosine schedule with warm-up ratio 0.03, HF's
    `get_cosine_schedule_with_warmup`; gradient accumulation 2; global-norm clipping 1.0; bf16 weights + fp32 master) over a pool of
    sixteen micro-batches (loss 10.6 -> ~3.1): HIP `Zero2AdamW` against the oracle trained the same way on the host (fp32 math on bf16-rounded working
    weights, fp32 master / moments, oracle.ref_ops.adamw_step).  Asserts the per-step relative gap and the final-loss gap."""
    import math
    from transformers import get_cosine_schedule_with_warmup
    from test_model_gpu import hip_model
    from oracle.ref_model import OracleConfig, forward as oracle_forward, init_state_dict
    from metamorph_amd.zero2 import Zero2AdamW
    V, START = 32002, 32000
    cfg = OracleConfig(hidden_size=1024, intermediate_size=2816, num_hidden_layers=4, num_attention_heads=8, num_key_value_heads=2,
                       vocab_size=V, rope_theta=10000.0, v_layers=2, v_intermediate=144, v_image=56, num_image_tokens=4,
                       tokenizer_model_max_length=256, image_start_id=START)
    seed, peak_lr, steps, accum = 9, 1e-4, 50, 2
    warm = math.ceil(steps * 0.03)
    pool = _curve_batches(16, seed=12, V=V, start_id=START)

    # ---- HIP
    model = hip_model(cfg, init_state_dict(cfg, seed=seed, dtype=torch.bfloat16, fast_big=True))
    model.train()
    opt = Zero2AdamW([p for p in model.parameters() if p.requires_grad], lr=peak_lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                     max_grad_norm=1.0)
    sched = get_cosine_schedule_with_warmup(opt, warm, steps)
    dev_pool = [{k: (v.cuda().bfloat16() if k == "images" else v.cuda()) for k, v in b.items()} for b in pool]
    hip_losses, hip_lr = [], []
    for s in range(steps):
        opt.zero_grad()
        tot = 0.0
        for a in range(accum):
            out = model(**dev_pool[(s * accum + a) % len(pool)])
            (out.loss / accum).backward()
            tot += float(out.loss.detach()) / accum
        hip_lr.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        hip_losses.append(tot)

    # ---- oracle
    master = init_state_dict(cfg, seed=seed, fast_big=True)
    train = [k for k in master if "vision_tower" not in k and "vision_proj" not in k]
    mom = {k: (torch.zeros_like(master[k]), torch.zeros_like(master[k])) for k in train}
    ora_losses = []
    for s in range(steps):
        lr = peak_lr * (s / max(1, warm) if s < warm else max(0.0, 0.5 * (1.0 + math.cos(math.pi * (s - warm) / max(1, steps - warm)))))
        assert abs(lr - hip_lr[s]) <= 1e-12 + 1e-9 * peak_lr, (s, lr, hip_lr[s])
        sd = {k: v.bfloat16().float() for k, v in master.items()}
        for k in train:
            sd[k].requires_grad_(True)
        tot = 0.0
        for a in range(accum):
            b = pool[(s * accum + a) % len(pool)]
            o = oracle_forward(sd, cfg, b["input_ids"], b["attention_mask"], b["labels"], b["images"].bfloat16().float(), return_logits=False,
                               ce_rows_only=True)
            (o["loss"] / accum).backward()
            tot += float(o["loss"].detach()) / accum
        ora_losses.append(tot)
        grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in train}
        norm = float(torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads.values())))
        coef = min(1.0, 1.0 / (norm + 1e-6))
        for k in train:
            R.adamw_step(master[k], grads[k], mom[k][0], mom[k][1], s + 1, lr, 0.9, 0.999, 1e-8, 0.0, grad_scale=coef)
    gaps = [abs(a - b) / max(abs(b), 1.0) for a, b in zip(hip_losses, ora_losses)]
    print("\n   long loss curve (every 5th step)  hip:", " ".join(f"{x:.4f}" for x in hip_losses[::5]), f"... {hip_losses[-1]:.4f}",
          "\n                                 oracle:", " ".join(f"{x:.4f}" for x in ora_losses[::5]), f"... {ora_losses[-1]:.4f}",
          f"\n   max per-step gap {max(gaps):.3e} (step {gaps.index(max(gaps))}), final gap {gaps[-1]:.3e}")
    assert ora_losses[-1] < ora_losses[0] - 1.0, ora_losses
    assert max(gaps) <= LC_STEP_TOL and gaps[-1] <= LC_FINAL_TOL, (max(gaps), gaps[-1])


LC_STEP_TOL, LC_FINAL_TOL = 1.4e-2, 3e-3                  # 1.5 x measured on MI355X (9.2e-3 at step 42, 1.8e-3 final)


def test_async_update_equals_synchronous_update():
    """The update kernels of step() run per segment on a side stream and the next forward pass waits segment by segment
    (functional.params_ready): four training steps of the tiny model must give the same parameters with and without it."""
    import os
    from conftest import GOLDEN
    from test_model_gpu import T, hip_model, tiny_cfg
    from oracle.ref_model import init_state_dict
    from metamorph_amd import functional as F
    from metamorph_amd.zero2 import Zero2AdamW, tag_segments
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    args = dict(input_ids=T(g["input_ids"]).cuda(), attention_mask=T(g["attention_mask"]).cuda(), labels=T(g["labels"]).cuda(),
                images=T(g["images"]).cuda().bfloat16())
    results = []
    for use_async in (False, True):
        cfg = tiny_cfg(num_image_tokens=4)
        model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
        model.train()
        tag_segments(model)
        opt = Zero2AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0, async_update=use_async)
        opt.enable_overlap()
        assert opt.async_update == use_async and len(opt.segs) == 2 + cfg.num_hidden_layers
        losses = []
        for _ in range(4):
            opt.zero_grad()
            out = model(**args)
            opt.arm_overlap()
            out.loss.backward()
            opt.step()
            if use_async:
                assert opt._ready                      # updates are in flight / recorded, nothing was waited for on the host
            losses.append(float(out.loss.detach()))
        opt.synchronize()
        results.append((losses, opt.flat_param.clone(), opt.master.clone()))
        F.set_layer_grad_hook(None)
        F.set_param_ready_hook(None)
    # (run-to-run the loss / gradient sums use fp32 atomics, so two runs agree to rounding, not bit for bit)
    for a, b in zip(results[0][0], results[1][0]):
        assert abs(a - b) <= 2e-4 * abs(a), (results[0][0], results[1][0])
    # Adam turns rounding-level gradient differences into +-lr steps on a few elements: compare the parameter vectors in norm
    d = (results[0][2] - results[1][2]).norm() / results[0][2].norm()
    assert float(d) <= 1e-3, float(d)
    assert float((results[0][1].float() - results[1][1].float()).norm() / results[0][1].float().norm()) <= 2e-3


def test_rccl_call_pattern_single_rank():
    """The collective call pattern of the multi-GPU path -- in-place reduce-scatter per segment launched asynchronously from
    DecoderLayerFn.backward, fp32 norm all-reduce, per-segment all-gather (synchronous and on the side stream) -- driven through
    real RCCL with a one-rank process group (zero2.set_collective_mode(force_collectives=True)): same losses as the collective-free run."""
    import os
    import socket
    import torch.distributed as dist
    from conftest import GOLDEN
    from test_model_gpu import T, hip_model, tiny_cfg
    from oracle.ref_model import init_state_dict
    from metamorph_amd import functional as F
    from metamorph_amd.zero2 import Zero2AdamW, tag_segments
    g = np.load(os.path.join(GOLDEN, "e2e_mixed_T4_ar1_bf16.npz"))
    args = dict(input_ids=T(g["input_ids"]).cuda(), attention_mask=T(g["attention_mask"]).cuda(), labels=T(g["labels"]).cuda(),
                images=T(g["images"]).cuda().bfloat16())

    def train(async_update):
        cfg = tiny_cfg(num_image_tokens=4)
        model = hip_model(cfg, init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16))
        model.train()
        tag_segments(model)
        opt = Zero2AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0, async_update=async_update)
        opt.enable_overlap()
        losses = []
        for _ in range(3):
            opt.zero_grad()
            out = model(**args)
            opt.arm_overlap()
            out.loss.backward()
            opt.step()
            losses.append(float(out.loss.detach()))
        opt.synchronize()
        F.set_layer_grad_hook(None)
        F.set_param_ready_hook(None)
        return losses, opt

    base, opt0 = train(False)
    assert not opt0._coll
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    _old_mode = __import__("metamorph_amd.zero2", fromlist=["x"]).set_collective_mode(force_collectives=True)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for async_update in (False, True):
            losses, opt = train(async_update)
            assert opt._coll and opt.world == 1
            for a, b in zip(losses, base):
                assert abs(a - b) <= 2e-4 * abs(b), (async_update, losses, base)
    finally:
        dist.destroy_process_group()
        __import__("metamorph_amd.zero2", fromlist=["x"]).set_collective_mode(**_old_mode)
