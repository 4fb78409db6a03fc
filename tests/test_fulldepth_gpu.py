"""FULL-DEPTH parity of the model `bench.py` times (BASELINE configs[1]): 32 LLaMA-3-8B decoder layers + 27 SO400M tower layers,
2048-token spliced samples, the weights `bench.build_bench_model` draws under the bench's seed -- against the fp32 oracle run
layer-streamed on the host cores of the GPU box (oracle/ref_stream.py: the per-layer functions pinned to the reference by the
golden vectors, weights read back from the device model one tensor at a time, so fp32 8B never sits in RAM).

Compared: tower output after 27 layers, hidden rows after 1 / 8 / 16 / 32 decoder layers, the final-norm hidden rows, loss /
loss_language / loss_image_ar (north_star: 1e-3), integer bookkeeping (bit-exact), and -- through the streamed backward chain --
the gradients of every tensor of decoder layers 0 and 31, the final norm, lm_head, vision_head and mm_projector.
Reference: metamorph_llama.py:349-359, 398-413, 420-474; siglip_encoder.py:138-163, 206-208.
"""
import json
import os
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.ref_model import OracleConfig  # noqa: E402
from oracle.ref_stream import full_depth  # noqa: E402
from test_model_gpu import ORACLE_FP32_DEVICE, RECORD_YARDSTICK, load_yardstick, oracle_fp32, record_yardstick  # noqa: E402

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
    # the fp32 oracle: host evaluation, or (default) the same functions through stock torch fp32 ops on the GPU -- tests/test_model_gpu.py
    # (ORACLE_FP32_DEVICE, test_streamed_oracle_on_device_*) -- weights read from the device model one tensor at a time either way
    ref = oracle_fp32(lambda k: sdict[k].detach().float().cpu(), lambda k: sdict[k].detach().float(), cfg, ids.cpu(), mask.cpu(), labels.cpu(),
                      images.float().cpu(), probe_layers=PROBES, grad_layers=(0, layers - 1), log=None)
    print(f"\n   [full depth {layers}+{vit_layers}] oracle ({'host' if ORACLE_FP32_DEVICE == 'cpu' or RECORD_YARDSTICK else ORACLE_FP32_DEVICE}) time "
          f"{time.time() - t0:.0f}s {({k: round(v, 1) for k, v in ref['seconds'].items()})} rows per sample {ref['n_rows']}")

    # the yardstick for the depth-accumulated error: the SAME streamed oracle run in bf16 on the host -- the reference stack's own bf16
    # arithmetic (HF modules in bf16 on the CPU) -- against its fp32 run, forward AND backward (the same 28 gradient tensors).  It is a
    # deterministic function of the seeded weights and batch (bit-identical in rounds 3, 4, 5 and 6): recorded in
    # tests/golden/r6_bf16_yardstick.json beside the fp32 oracle's loss, which is re-checked here; MM355_RECORD_YARDSTICK=1 (or a missing
    # entry, or other layer counts) computes it on the spot (68 s of host time)
    yard = load_yardstick("configs1_full_depth") if not full else None
    if yard is None and os.environ.get("MM355_FULLDEPTH_REF_BF16", "1") != "0":
        t0 = time.time()
        ref16 = full_depth(lambda k: sdict[k].detach().cpu(), cfg, ids.cpu(), mask.cpu(), labels.cpu(), images.cpu(),
                           probe_layers=PROBES, grad_layers=(0, layers - 1), backward=True)
        va = ref["attention_mask"]
        yard = dict(oracle_fp32_loss=ref["loss"], rows=ref["n_rows"], tower=rel(ref16["raw_hidden"], ref["raw_hidden"]),
                    hidden_after_layers={str(n): rel(ref16["probes"][n][va], ref["probes"][n][va]) for n in PROBES if n <= layers},
                    final_norm=rel(ref16["hidden_states"][va], ref["hidden_states"][va]), loss=ref16["loss"], loss_language=ref16["loss_language"],
                    loss_image_ar=ref16["loss_image_ar"], grads={k: rel(ref16["grads"][k], g) for k, g in ref["grads"].items()})
        print(f"   oracle in bf16 on the host: {time.time() - t0:.0f}s")
        if RECORD_YARDSTICK and not full:
            record_yardstick("configs1_full_depth", yard)
    elif yard is not None:
        assert abs(ref["loss"] - yard["oracle_fp32_loss"]) <= 2e-5 * abs(ref["loss"]) and ref["n_rows"] == yard["rows"], (
            f"the fp32 oracle gives loss {ref['loss']} on rows {ref['n_rows']}; the recorded bf16 yardstick was taken beside "
            f"{yard['oracle_fp32_loss']} / {yard['rows']} -- re-record with MM355_RECORD_YARDSTICK=1")
    e16 = None
    if yard is not None:
        e16 = {n: yard["hidden_after_layers"][str(n)] for n in PROBES if n <= layers}
        print(f"   oracle in bf16 vs oracle in fp32: tower {yard['tower']:.3e}  hidden after n layers "
              + "  ".join(f"{n}: {e:.3e}" for n, e in e16.items()) + f"  final norm {yard['final_norm']:.3e}"
              + f"  loss {yard['loss']:.5f} (lang {yard['loss_language']:.5f} img {yard['loss_image_ar']:.5f})")

    # ---- measure everything first (and leave the numbers behind even if an assert below fires), then judge
    valid = ref["attention_mask"]
    e_raw = rel(raw, ref["raw_hidden"])
    errs = {n: rel(taps[n].view(2, 2048, -1)[valid], ref["probes"][n][valid]) for n in PROBES if n <= layers}
    e_fin = rel(out.hidden_states.float().cpu()[valid], ref["hidden_states"][valid])
    params = dict(model.named_parameters())
    worst, per_tensor, per_tensor16 = {}, {}, {}
    for k, g in ref["grads"].items():
        e = rel(params[k].grad, g) if params[k].grad is not None else float("inf")
        per_tensor[k] = e
        if yard is not None and k in yard["grads"]:
            per_tensor16[k] = yard["grads"][k]
        grp = "layer " + k.split(".")[2] if k.startswith("model.layers.") else "heads/projector"
        worst[grp] = max(worst.get(grp, (0.0, "")), (e, k))
    print(f"   tower hidden_states[-1] after {vit_layers} layers: rel err {e_raw:.3e}")
    print("   hidden rows rel err after n decoder layers: " + "  ".join(f"{n}: {e:.3e}" for n, e in errs.items()) + f"  final norm: {e_fin:.3e}")
    print(f"   loss hip={got_loss:.5f} oracle-fp32={ref['loss']:.5f}  lang {got_lang:.5f}/{ref['loss_language']:.5f}  "
          f"img {got_img:.5f}/{ref['loss_image_ar']:.5f}")
    print("   gradients, worst rel err: " + "  ".join(f"{g}: {e:.3e} ({k.split('.', 3)[-1] if g != 'heads/projector' else k})" for g, (e, k) in worst.items()))
    if per_tensor16:
        print("   gradients vs fp32, HIP | the oracle in bf16 (the reference stack's own arithmetic):")
        for k in sorted(per_tensor, key=lambda k_: -per_tensor[k_]):
            print(f"      {k}: {per_tensor[k]:.3e} | {per_tensor16.get(k, float('nan')):.3e}")
    record = dict(layers=layers, tower_layers=vit_layers, rows=ref["n_rows"], oracle_seconds=ref["seconds"], tower_rel_err=e_raw,
                  hidden_rel_err_after_layers=errs, final_norm_rel_err=e_fin, loss=dict(hip=got_loss, oracle=ref["loss"]),
                  loss_language=dict(hip=got_lang, oracle=ref["loss_language"]), loss_image_ar=dict(hip=got_img, oracle=ref["loss_image_ar"]),
                  grad_rel_err=per_tensor, grad_rel_err_oracle_bf16=per_tensor16)
    if yard is not None:
        record["oracle_bf16_vs_fp32"] = dict(tower=yard["tower"], hidden_after_layers=e16, final_norm=yard["final_norm"], loss=yard["loss"],
                                             loss_language=yard["loss_language"], loss_image_ar=yard["loss_image_ar"])
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "fulldepth_parity.json"), "w") as f:
            json.dump(record, f, indent=1)
    except OSError:
        pass

    # integer bookkeeping of the splice: bit-exact
    assert torch.equal(plan[5].cpu(), ref["labels"]) and torch.equal(plan[6].cpu(), ref["image_positions"])
    assert torch.equal(plan[2].cpu().bool(), ref["attention_mask"])
    assert ref["n_rows"] == [GEN_IDS + 255, 2048]
    # losses: north_star's 1e-3
    assert abs(got_loss - ref["loss"]) <= 1e-3 * abs(ref["loss"])
    assert abs(got_lang - ref["loss_language"]) <= 1e-3 * abs(ref["loss_language"])
    assert abs(got_img - ref["loss_image_ar"]) <= 1e-3
    # bf16 rounding accumulates over depth (2-layer models: 8e-3).  Two bounds: absolute = 1.5 x the value measured on MI355X in
    # round 3 (profiles/r3_fulldepth_parity.log), and relative to the reference stack's own bf16 arithmetic at the same depth
    # (measured: the HIP path is 6-10 % CLOSER to the fp32 truth than the bf16 oracle at every probe)
    assert e_raw <= TOWER_TOL, e_raw
    for n, e in errs.items():
        assert e <= HIDDEN_TOL[n], (n, e)
    assert e_fin <= HIDDEN_TOL["final"], e_fin
    if yard is not None:
        assert e_raw <= 1.15 * record["oracle_bf16_vs_fp32"]["tower"], (e_raw, record["oracle_bf16_vs_fp32"]["tower"])
        for n, e in errs.items():
            assert e <= 1.15 * e16[n], (n, e, e16[n])
        assert e_fin <= 1.15 * record["oracle_bf16_vs_fp32"]["final_norm"]
        assert abs(got_loss - ref["loss"]) <= max(3.0 * abs(yard["loss"] - ref["loss"]), 2e-4 * abs(ref["loss"]))
    assert len(ref["grads"]) == 18 + 2 + 4 + 4
    for k, e in per_tensor.items():
        if per_tensor16:
            # the yardstick: the reference stack's own bf16 run of the same backward chain; each HIP gradient within 1.5 x its distance from
            # the fp32 truth (floor: bf16 resolution of a depth-32 chain, the 3.3e-2 of tests/test_model_gpu.py's 2-layer goldens)
            assert e <= max(1.5 * per_tensor16[k], GRAD_FLOOR), (k, e, per_tensor16[k])
        else:
            # (MM355_FULLDEPTH_REF_BF16=0: no yardstick) q / k projections of a random-weight model receive near-noise gradients:
            # measured 9.6e-2 in layer 31, 6.2e-2 in layer 0; everything else <= 5.7e-2
            assert e <= (GRAD_TOL_QK if ("q_proj" in k or "k_proj" in k) else GRAD_TOL), (k, e)


TOWER_TOL = 1.8e-2                                               # measured 1.18e-2 (oracle in bf16: 1.31e-2)
HIDDEN_TOL = {1: 1.4e-2, 8: 3.0e-2, 16: 4.2e-2, 32: 5.9e-2, "final": 5.9e-2}     # measured 9.0e-3 / 2.0e-2 / 2.8e-2 / 3.9e-2 / 3.9e-2
GRAD_TOL, GRAD_TOL_QK = 8.5e-2, 1.45e-1
GRAD_FLOOR = 3.3e-2
