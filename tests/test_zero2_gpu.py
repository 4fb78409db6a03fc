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
