"""The layer-streamed oracle (oracle/ref_stream.py, used by the full-depth GPU parity test) against the plain oracle
(oracle/ref_model.forward, itself pinned to the reference by test_oracle_vs_golden.py) on a model small enough for both."""
import numpy as np
import torch

from oracle.ref_model import OracleConfig, forward as oracle_forward, init_state_dict
from oracle.ref_stream import full_depth


def _batch():
    A, ST, IM, EN, PAD = 128000, 128256, -200, 128257, 128001
    g = torch.Generator().manual_seed(2)
    ids = torch.full((2, 30), PAD, dtype=torch.long)
    a = torch.randint(0, 127000, (30,), generator=g)
    a[0] = a[1] = A
    a[5], a[6], a[7] = ST, IM, EN                                # prompt-side image, long sample
    ids[0] = a
    b = torch.randint(0, 127000, (14,), generator=g)
    b[0] = b[1] = A
    b[10], b[11], b[12], b[13] = ST, IM, EN, 128009              # answer-side image, short sample (padded)
    ids[1, :14] = b
    lab = torch.full_like(ids, -100)
    lab[0, 12:] = ids[0, 12:]
    lab[1, 6:14] = ids[1, 6:14]
    lab[1, 11] = IM
    images = torch.from_numpy(np.random.default_rng(4).standard_normal((2, 3, 56, 56), dtype=np.float32))
    return ids, ids.ne(PAD), lab, images


def test_streamed_oracle_equals_plain_oracle():
    cfg = OracleConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       vocab_size=128258, v_layers=2, v_intermediate=144, v_image=56, num_image_tokens=4, tokenizer_model_max_length=64)
    sd = init_state_dict(cfg, seed=3)
    ids, msk, lab, images = _batch()
    for k, v in sd.items():
        v.requires_grad_("vision_tower" not in k and "vision_proj" not in k)
    ref = oracle_forward(sd, cfg, ids, msk, lab, images, return_logits=False)
    ref["loss"].backward()
    got = full_depth(lambda k: sd[k].detach().clone(), cfg, ids, msk, lab, images, probe_layers=(1, 4), grad_layers=(0, 3), embed_grad=True)
    assert torch.equal(got["labels"], ref["labels"]) and torch.equal(got["image_positions"], ref["image_positions"])
    assert abs(got["loss"] - float(ref["loss"].detach())) <= 2e-6 * abs(got["loss"])
    assert abs(got["loss_language"] - ref["loss_language"]) <= 2e-6 * abs(ref["loss_language"])
    assert abs(got["loss_image_ar"] - ref["loss_image_ar"]) <= 2e-6
    valid = ref["attention_mask"]
    a, b = got["hidden_states"][valid], ref["hidden_states"].detach()[valid]
    assert float((a - b).norm() / b.norm()) < 1e-5
    assert set(got["probes"]) == {1, 4}
    n = 0
    for k, g in got["grads"].items():
        r = sd[k].grad
        assert r is not None, k
        assert float((g - r).norm() / r.norm().clamp_min(1e-20)) < 2e-4, k
        n += 1
    assert n == 2 * 9 + 2 + 4 + 4 + 1                            # two layers, norm + lm_head, vision_head, mm_projector, embed_tokens
    # layers outside grad_layers carry no gradient; forward-only mode returns none at all
    assert not any(k.startswith("model.layers.1.") for k in got["grads"])
    fwd = full_depth(lambda k: sd[k].detach().clone(), cfg, ids, msk, lab, images, backward=False)
    assert fwd["grads"] == {} and abs(fwd["loss"] - got["loss"]) < 1e-7


def test_streamed_oracle_against_reference_recorded_8_layer_run():
    """oracle/ref_stream.full_depth DIRECTLY against the reference: tests/golden/e2e_multi_frame_T4_ar1_deep8_f32.npz is the reference's own
    forward + backward of an 8-layer decoder (4 heads over one KV head) + 4-layer tower on the multi-frame batch (oracle/gen_golden.py e2e).
    Loss, final hidden rows and the gradient summaries of decoder layers 0 and 7, the heads and the projector."""
    import json
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "e2e_multi_frame_T4_ar1_deep8_f32.npz"))
    cfg = OracleConfig(hidden_size=256, intermediate_size=512, num_key_value_heads=1, vocab_size=128258, v_intermediate=144, v_image=56,
                       num_image_tokens=4, tokenizer_model_max_length=64, **json.loads(str(g["cfg_json"])))
    assert cfg.num_hidden_layers == 8 and cfg.v_layers == 4
    sd = init_state_dict(cfg, seed=int(g["seed"]))
    T = torch.from_numpy
    got = full_depth(lambda k: sd[k].detach().clone(), cfg, T(g["input_ids"]), T(g["attention_mask"]), T(g["labels"]), T(g["images"]),
                     grad_layers=(0, 7))
    assert abs(got["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    assert abs(got["loss_language"] - float(g["loss_language"])) <= 2e-5 * abs(float(g["loss_language"]))
    assert abs(got["loss_image_ar"] - float(g["loss_image_ar"])) <= 2e-5
    valid = got["attention_mask"]
    torch.testing.assert_close(got["hidden_states"][valid], T(g["hidden"])[valid], rtol=2e-4, atol=2e-5)

    def summary(t):
        f = t.detach().float().flatten()
        n = min(256, f.numel())
        idx = (torch.arange(n, dtype=torch.long) * (f.numel() - 1)) // max(n - 1, 1)
        return torch.cat([f.norm()[None], f[idx]])
    checked = 0
    for name, grad in got["grads"].items():
        key = "grad::" + name
        assert key in g.files, name
        torch.testing.assert_close(summary(grad), T(g[key]), rtol=5e-4, atol=2e-6)
        checked += 1
    assert checked == 18 + 2 + 4 + 4
