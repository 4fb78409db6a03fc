"""FULL-DEPTH parity of the model `bench.py` times (BASELINE configs[1]): 32 LLaMA-3-8B decoder layers + 27 SO400M tower layers,
2048-token spliced samples, the weights `bench.build_bench_model` draws under the bench's seed -- against the fp32 oracle run
layer-streamed on the host cores of the GPU box (oracle/ref_stream.py: the per-layer functions pinned to the reference by the
golden vectors, weights read back from the device model one tensor at a time, so fp32 8B never sits in RAM).

Compared: tower output after 27 layers, hidden rows after 1 / 8 / 16 / 32 decoder layers, the final-norm hidden rows, loss /
loss_language / loss_image_ar (north_star: 1e-3), integer bookkeeping (bit-exact), and -- through the streamed backward chain --
the gradients of every tensor of decoder layers 0 and 31, the final norm, lm_head, vision_head and mm_projector.
Reference: metamorph_llama.py:349-359, 398-413, 420-474; siglip_encoder.py:138-163, 206-208.
"""
import os
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_model import OracleConfig  # noqa: E402
from oracle.ref_stream import full_depth  # noqa: E402

DEV = "cuda"
GEN_IDS = 120            # ids kept of the generation sample (sample 0 of the bench batch): 120 + 255 = 375 spliced rows
PROBES = (1, 8, 16, 32)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_configs1_full_depth_against_streamed_oracle():
    import bench
    full = os.environ.get("MM355_FULLDEPTH_LAYERS")                  # debugging aid: fewer layers (NOT the claim of this test)
    layers, vit_layers = (int(full), min(int(full), 27)) if full else (32, 27)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 128)))
    model = bench.build_bench_model(torch.device(DEV), layers=layers, vit_layers=vit_layers, image_tokens=256)
    # the bench's own rank-0 batch generator (seed 1234): sample 0 = generation sample, sample 1 = image-QA sample of exactly
    # 2048 spliced rows; the generation sample is cut to GEN_IDS ids (padding after it) to bound the oracle's host time
    ids, labels, mask, images = bench.make_batch(2, 2048, 256, torch.device(DEV), seed=1234)
    ids[0, GEN_IDS:] = 128001
    labels[0, GEN_IDS:] = -100
    mask[0, GEN_IDS:] = False

    taps = {}
    model.get_model().layer_output_hook = lambda li, x: taps.__setitem__(li + 1, x.detach().float().cpu()) if (li + 1) in PROBES else None
    tower = model.get_model().vision_tower
    with torch.no_grad():
        raw = tower.vision_tower.forward_features(images, tower.select_layer).float().cpu()
    out = model(input_ids=ids, attention_mask=mask, labels=labels, images=images)
    model.get_model().layer_output_hook = None
    plan = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, labels, images)
    out.loss.backward()
    torch.cuda.synchronize()
    got_loss, got_lang, got_img = float(out.loss.detach()), model.loss_language, model.loss_image_ar

    cfg = OracleConfig(num_hidden_layers=layers, v_layers=vit_layers, num_image_tokens=256, tokenizer_model_max_length=4096)
    sdict = model.state_dict()
    t0 = time.time()
    ref = full_depth(lambda k: sdict[k].detach().float().cpu(), cfg, ids.cpu(), mask.cpu(), labels.cpu(), images.float().cpu(),
                     probe_layers=PROBES, grad_layers=(0, layers - 1), log=None)
    print(f"\n   [full depth {layers}+{vit_layers}] oracle host time {time.time() - t0:.0f}s {({k: round(v, 1) for k, v in ref['seconds'].items()})}"
          f" rows per sample {ref['n_rows']}")

    # integer bookkeeping of the splice: bit-exact
    assert torch.equal(plan[5].cpu(), ref["labels"]) and torch.equal(plan[6].cpu(), ref["image_positions"])
    assert torch.equal(plan[2].cpu().bool(), ref["attention_mask"])
    valid = ref["attention_mask"]
    assert ref["n_rows"] == [GEN_IDS + 255, 2048]

    # tower: 27 layers of bf16 against fp32
    e_raw = rel(raw, ref["raw_hidden"])
    print(f"   tower hidden_states[-1] after {vit_layers} layers: rel err {e_raw:.3e}")
    assert e_raw <= 2.5e-2

    # decoder: error per depth
    errs = {}
    for n in PROBES:
        if n > layers:
            continue
        hip = taps[n].view(2, 2048, -1)
        errs[n] = rel(hip[valid], ref["probes"][n][valid])
    e_fin = rel(out.hidden_states.float().cpu()[valid], ref["hidden_states"][valid])
    print("   hidden rows rel err after n decoder layers: " + "  ".join(f"{n}: {e:.3e}" for n, e in errs.items()) + f"  final norm: {e_fin:.3e}")
    for n, e in errs.items():
        assert e <= 3.5e-2, (n, e)
    assert e_fin <= 3.5e-2

    # losses: north_star's 1e-3
    print(f"   loss hip={got_loss:.5f} oracle-fp32={ref['loss']:.5f}  lang {got_lang:.5f}/{ref['loss_language']:.5f}  "
          f"img {got_img:.5f}/{ref['loss_image_ar']:.5f}")
    assert abs(got_loss - ref["loss"]) <= 1e-3 * abs(ref["loss"])
    assert abs(got_lang - ref["loss_language"]) <= 1e-3 * abs(ref["loss_language"])
    assert abs(got_img - ref["loss_image_ar"]) <= 1e-3

    # gradients through the full chain: first and last decoder layer, heads, projector
    params = dict(model.named_parameters())
    worst = {}
    for k, g in ref["grads"].items():
        assert params[k].grad is not None, k
        e = rel(params[k].grad, g)
        grp = k.split(".")[2] if k.startswith("model.layers.") else "heads"
        worst[grp] = max(worst.get(grp, (0.0, "")), (e, k))
    print("   gradients, worst rel err: " + "  ".join(f"layer {g}: {e:.3e} ({k.split('.', 3)[-1]})" if g != "heads" else f"heads/projector: {e:.3e} ({k})"
                                                      for g, (e, k) in worst.items()))
    assert len(ref["grads"]) == 18 + 2 + 4 + 4
    for g, (e, k) in worst.items():
        assert e <= 8e-2, (k, e)
