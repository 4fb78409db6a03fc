#!/usr/bin/env python3
"""Headline benchmark: MetaMorph instruction-tuning step throughput on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LLaMA-3-8B geometry + SigLIP-SO400M/14-384 tower (the tower the reference
hard-codes), 2048-token spliced sequences with one 256-token image each, bf16, full fine-tune of everything
but the tower (reference stage 2), fused AdamW with ZeRO-2 sharding over RCCL.  One step = zero_grad + forward +
backward + gradient reduce-scatter + AdamW shard update + parameter all-gather on one batch of synthetic samples;
weights are random (no checkpoints can be downloaded), data is synthetic and resident in HBM.

Prints ONE JSON line (rank 0).  `value` = total valid spliced tokens per second over all GPUs.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch
import torch.distributed as dist

MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md (256 CU x 2.4 GHz x 4096 FLOP/clk/CU)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="samples per GPU per step (each 2048 spliced tokens); 16 = 222 GB of the 288 GB HBM")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--image-tokens", type=int, default=256)
    ap.add_argument("--train-vision", action="store_true", help="freeze_vision=False: the SigLIP tower trains too, own lr group (SURVEY row N4)")
    ap.add_argument("--all-generation", action="store_true", help="every sample is a generation sample (BASELINE configs[3], a parity-test case)")
    ap.add_argument("--frames", type=int, default=1, help="images per sample (8 with --seq 4096 = BASELINE configs[2], a parity-test case)")
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer decoder layers => NOT the headline config")
    ap.add_argument("--vit-layers", type=int, default=27)
    ap.add_argument("--zero", type=int, default=2, choices=(2, 3), help="3: decoder-layer parameters sharded (Zero3AdamW, BASELINE configs[4] machinery); NOT the headline config")
    ap.add_argument("--grad-checkpointing", action="store_true", help="per-layer recompute (reference --gradient_checkpointing True); NOT the headline config")
    ap.add_argument("--set-variant", action="append", default=[], metavar="NAME=0|1",
                    help="flip a composition switch of metamorph_amd.functional.VARIANTS for a same-box A/B (tools; the default line uses none)")
    ap.add_argument("--ckpt-layers", default="all", help="with --grad-checkpointing: 'all' (the reference's behaviour), an integer n (only the first n "
                    "decoder layers recompute, the others keep their activations), or 'auto' (the fewest layers that fit this GPU's HBM)")
    ap.add_argument("--host-inputs", action="store_true", help="ids / labels / mask / fp32 pixels start every step in pinned HOST memory (PCIe-inclusive "
                    "rate for DESIGN.md; never the headline `value`, whose inputs are resident in HBM)")
    ap.add_argument("--zero2-async", type=int, default=None, choices=(0, 1), help="1: AdamW shard update + parameter all-gather per segment on a side "
                    "stream, the next forward waits segment by segment (hides the all-gather at world > 1); default 0")
    ap.add_argument("--zero2-overlap", type=int, default=1, choices=(0, 1), help="0: the gradient reduce-scatters all start at step() instead of from "
                    "inside backward (driver-side A/B of the overlap; default 1)")
    ap.add_argument("--pool", type=int, default=8, help="distinct seeded batches rotated through the steps (the loss is then a training loss, not a memorised batch)")
    ap.add_argument("--llama31-rope", action="store_true", help="LLaMA-3.1's rope_scaling (rope_type llama3) instead of the plain theta = 5e5 RoPE of "
                    "BASELINE's LLaMA-3-8B: only the cos / sin TABLES differ, every kernel is the same; said in config.workload")
    ap.add_argument("--ragged", type=float, default=0.0, metavar="SHARE", help="ragged batches with this mean padding share (sample lengths uniform in "
                    "[1 - 2 SHARE, 1] x seq, right-padded to the longest): `value` then counts VALID tokens/s; NOT the headline config")
    ap.add_argument("--compact-rows", default="auto", choices=("auto", "on", "off", "exact"), help="padding-free decoder rows (model.config.mm355_compact_rows): "
                    "auto = from 8 %% padding")
    ap.add_argument("--alloc-roundup", type=int, default=0, metavar="DIV", help="experiment: torch caching allocator roundup_power2_divisions (sizes that "
                    "change from step to step under --compact-rows exact then fall into few block classes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def make_batch(B, L, T, device, seed, frames=1, all_generation=False, ragged=0.0):
    """[BOS,BOS, 20 text, frames x (<image_start>, <image>, <image_end>), text ...] padded so the SPLICED length is exactly L.
    Samples 1..B-1 are image-QA (prompt-side images, labels -100 on the prompt); sample 0 is a generation sample (its LAST
    image is answer-side: the label at its <image_start> is live) so the vision-head / cosine path runs and the combined loss is
    finite (with no answer image the reference's loss is NaN).  frames=8, L=4096 is BASELINE configs[2]."""
    g = torch.Generator().manual_seed(seed)
    n_ids = L - frames * (T - 1)
    ids = torch.randint(0, 127999, (B, n_ids), generator=g)
    ids[:, 0] = 128000
    ids[:, 1] = 128000
    img_pos = [22 + 3 * f + 1 for f in range(frames)]         # positions of the -200 markers
    for p in img_pos:
        ids[:, p - 1], ids[:, p], ids[:, p + 1] = 128256, -200, 128257
    labels = ids.clone()
    prompt_end = img_pos[-1] + 2 + 20                         # images + 20 more prompt tokens are never supervised
    labels[:, :prompt_end] = -100
    for p in img_pos:
        labels[:, p] = -100
    last = img_pos[-1]
    gen = slice(0, B) if all_generation else slice(0, 1)     # BASELINE configs[3]: every sample regresses its answer image
    labels[gen, last - 3:] = ids[gen, last - 3:]             # generation sample: supervise from just before the last <image_start>
    labels[gen, last] = -200
    mask = torch.ones(B, n_ids, dtype=torch.bool)
    if ragged > 0:
        # --ragged SHARE: an ImageQA-like length distribution -- sample lengths uniform in [1 - 2 SHARE, 1] x the full length (mean padding share
        # SHARE), right-padded to the longest as the reference's collator does (train.py:1258-1284); one sample per batch keeps the full length
        u = 1.0 - 2.0 * ragged * torch.rand(B, generator=g)
        u[B - 1] = 1.0
        for b in range(B):
            nb = max(prompt_end + 8, int(round(float(u[b]) * n_ids)))
            ids[b, nb:], labels[b, nb:], mask[b, nb:] = 128001, -100, False
    images = torch.randn(B * frames, 3, 384, 384, generator=g)
    if device is None:                                       # --host-inputs: what a DataLoader with pin_memory hands the Trainer
        return tuple(t.pin_memory() for t in (ids, labels, mask, images))
    # resident in HBM before the timed region; the integer tensors keep their host originals as mirrors -- what the collator of a real run
    # holds anyway (reference batch contract train.py:1258-1284) -- so that the splice plan is built without a device -> host copy
    from metamorph_amd import hostmirror
    return (hostmirror.to_device(ids, device), hostmirror.to_device(labels, device), hostmirror.to_device(mask, device),
            images.to(device).to(torch.bfloat16))


LLAMA31_ROPE = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)


def build_bench_model(dev, layers=32, vit_layers=27, image_tokens=256, seed=1234, llama31_rope=False):
    """The model this benchmark times: LLaMA-3-8B geometry + SO400M tower, random weights from torch's device generator under a
    fixed seed (identical on every rank; the BATCH is seeded per rank).  tests/test_fulldepth_gpu.py builds the same model with the
    same seed and compares its step-0 forward / backward with the fp32 oracle at full depth."""
    from metamorph_amd.factory import LLAMA3_8B, build_model
    llm = dict(LLAMA3_8B, num_hidden_layers=layers)
    if llama31_rope:                                             # meta-llama/Llama-3.1-8B's config.json (the reference README's base model)
        llm.update(rope_scaling=dict(LLAMA31_ROPE), max_position_embeddings=131072)
    geo = dict(num_hidden_layers=vit_layers)
    torch.manual_seed(seed)
    model = build_model(llm, geo, num_image_tokens=image_tokens, max_length=4096, device=dev, init_on_device=True)
    model.train()
    return model


class GemmTimer:
    """HIP-event timing of every GEMM launch (torch.cuda.Event == hipEvent on the stream the kernels are launched on)."""

    def __init__(self):
        from metamorph_amd import ops
        self.ops = ops
        self.orig = ops.gemm
        self.records = []
        self.enabled = False

    def install(self):
        def timed(a, b, out=None, **kw):
            if not self.enabled:
                return self.orig(a, b, out, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig(a, b, out, **kw)
            e.record()
            M, K = a.shape
            N = kw.get("n") or b.shape[0]
            self.records.append((s, e, 2.0 * M * N * K, (M, N, K), 2.0 * (M * K + N * K + M * N)))
            return r
        self.ops.gemm = timed
        import metamorph_amd.functional as F
        F.ops.gemm = timed
        orig_pair = self.ops.gemm_pair

        def timed_pair(a0, b0, out0, acc0, a1, b1, out1, acc1):     # two problems, one launch: flops of both, one duration
            if not self.enabled:
                return orig_pair(a0, b0, out0, acc0, a1, b1, out1, acc1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_pair(a0, b0, out0, acc0, a1, b1, out1, acc1)
            e.record()
            (m0, k0), n0, (m1, k1), n1 = a0.shape, b0.shape[0], a1.shape, b1.shape[0]
            self.records.append((s, e, 2.0 * (m0 * n0 * k0 + m1 * n1 * k1), ("pair", m0, n0, k0, m1, n1, k1),
                                 2.0 * (m0 * k0 + n0 * k0 + m0 * n0 + m1 * k1 + n1 * k1 + m1 * n1)))
            return r
        self.ops.gemm_pair = timed_pair
        # the fused MLP launches are GEMMs too (gate|up + SwiGLU; down_proj input gradient + SwiGLU backward): their flops over their WHOLE
        # duration, fused epilogue included
        orig_sw, orig_swb = self.ops.gemm_swiglu, self.ops.gemm_swiglu_bwd

        def timed_sw(x, wgu, I):
            if not self.enabled:
                return orig_sw(x, wgu, I)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_sw(x, wgu, I)
            e.record()
            M, K = x.shape
            self.records.append((s, e, 2.0 * M * 2 * I * K, ("swiglu", M, 2 * I, K), 2.0 * (M * K + 2 * I * K + M * 2 * I)))
            return r

        def timed_swb(dy, wdT, gu, I):
            if not self.enabled:
                return orig_swb(dy, wdT, gu, I)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_swb(dy, wdT, gu, I)
            e.record()
            M, K = dy.shape
            self.records.append((s, e, 2.0 * M * I * K, ("swiglu_bwd", M, I, K), 2.0 * (M * K + I * K + M * I)))
            return r
        self.ops.gemm_swiglu, self.ops.gemm_swiglu_bwd = timed_sw, timed_swb
        orig_rope = self.ops.gemm_rope                       # q|k|v GEMM with the RoPE rotation in its epilogue: a plain GEMM launch (+ 1 %)

        def timed_rope(x, wqkv, *a, **kw):
            if not self.enabled:
                return orig_rope(x, wqkv, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_rope(x, wqkv, *a, **kw)
            e.record()
            M, K = x.shape
            N = wqkv.shape[0]
            self.records.append((s, e, 2.0 * M * N * K, (M, N, K), 2.0 * (M * K + N * K + M * N)))
            return r
        self.ops.gemm_rope = timed_rope

    def fused_summary(self):
        """(launches, seconds, GEMM flops) of the fused MLP launches (gate|up + SwiGLU; down_proj input gradient + SwiGLU backward)."""
        recs = [r for r in self.records if r[3][0] in ("swiglu", "swiglu_bwd")]
        return len(recs), sum(r[0].elapsed_time(r[1]) for r in recs) * 1e-3, sum(r[2] for r in recs)

    def summary(self, plain_only=False):
        torch.cuda.synchronize()
        recs = [r for r in self.records if not (plain_only and r[3][0] in ("swiglu", "swiglu_bwd"))]
        t = sum(r[0].elapsed_time(r[1]) for r in recs) * 1e-3
        fl = sum(r[2] for r in recs)
        if plain_only:
            return len(recs), t, fl, sum(r[4] for r in recs)
        if os.environ.get("MM355_BENCH_GEMM_TABLE") == "1":   # per-shape breakdown on stderr (tuning aid)
            by = {}
            for s, e, f, shp, _ in self.records:
                c = by.setdefault(shp, [0, 0.0, 0.0])
                c[0] += 1; c[1] += s.elapsed_time(e) * 1e-3; c[2] += f
            for shp, (n, tt, ff) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                print(f"[gemm] M,N,K={shp} calls={n} total={tt * 1e3:8.2f} ms  {ff / tt / 1e12:7.1f} TF/s", file=sys.stderr)
        return len(self.records), t, fl


class HbmTimer:
    """HIP-event timing of the step's HBM-bound kernels (RMSNorm, transposes, AdamW): algorithmic bytes over launch duration."""

    def __init__(self):
        from metamorph_amd import ops
        self.ops = ops
        self.records = {}
        self.enabled = False

    def _wrap(self, name, nbytes):
        orig = getattr(self.ops, name)

        def timed(*a, **kw):
            if not self.enabled:
                return orig(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **kw)
            e.record()
            self.records.setdefault(name, []).append((s, e, float(nbytes(*a, **kw))))
            return r
        setattr(self.ops, name, timed)                        # functional.py / zero2.py reach the kernels through this module object

    def install(self):
        nb = lambda t: t.numel() * t.element_size()
        self._wrap("rmsnorm_fwd", lambda x, w, eps, out=None, want_rstd=False: 2 * nb(x) + nb(w))                    # 2 M h 2 B (+ h 2 B)
        self._wrap("rmsnorm_apply_t", lambda x, w, rstd, Rp=None: 2 * nb(x) + nb(w) + nb(rstd))
        self._wrap("transpose", lambda x, out=None, ld_out=None: 2 * nb(x))                                         # 2 R C 2 B
        self._wrap("adamw_shard_", lambda p32, m, v, g, p_out, *a, **kw: 2 * (nb(p32) + nb(m) + nb(v)) + nb(g) + nb(p_out))   # 28 B / parameter

    def summary(self, steps):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            t = sum(r[0].elapsed_time(r[1]) for r in recs) * 1e-3
            by = sum(r[2] for r in recs)
            out[name.rstrip("_")] = {"launches_per_step": round(len(recs) / steps, 1), "ms_per_step": round(t / steps * 1e3, 2),
                                     "achieved_tb_s": round(by / t / 1e12, 2), "frac_of_8_tb_s": round(by / t / 8e12, 3)}
        return out


def _cpu_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 32))          # 32 threads beat 64 / 128 / 256 on the GPU box host (profiles/r3_host_threads.log)


# ------------------------------------------------------------------------------------------------ pre-registered N-GPU expectation
# This build has never run on more than one GPU (every lease is a 1-GPU box).  So that the driver's 1 / 2 / 4 / 8-GPU run is a TEST and not
# a first look, the expectation is written down here and in DESIGN.md section 6 BEFORE that run, from one-GPU measurements + stated link
# assumptions, and every multi-rank bench line carries it (`predicted`) beside what it measured.
ONE_RANK_REFERENCE_MS = {(16, 2048, 1): 1240.0}      # (per-GPU batch, seq, frames) -> one-rank ms/step (profiles/r5_bench_default_*.json: 1 233 - 1 248)
XGMI_LINK_GBS = 153.0            # per link and direction, MI355X_MICROARCH.md (7 links per GPU, point-to-point)
XGMI_EFFICIENCY = 0.8            # of the link rate that RCCL's kernels sustain (assumed)
GEMM_SLOWDOWN_UNDER_COLLECTIVE = 0.10    # measured on one GPU with a CU-holding side kernel: 7-13 % (profiles/r2_gemm_cu_thief.log)


def predict_step(world, step1_ms, adamw_ms, seg_bytes, tail_bytes, gemm_share=0.83, layer_bwd_ms=None, n_layers=32):
    """Per-rank step time of the weak-scaling job at `world` ranks from the one-rank step (`step1_ms`, of which `adamw_ms` is the sharded
    update): gradients of one decoder layer (`seg_bytes`, bf16) are reduce-scattered while the backward of the next layer runs, the
    `tail_bytes` of everything outside the decoder (embeddings, lm_head, heads) at step(), all parameters all-gathered after the update.
    Two collective algorithms bound the prediction: "ring" (every byte crosses ONE link per hop: (N-1)/N * S per link) and "direct" (shard j
    goes straight to rank j over its own link: S / N per link)."""
    if world <= 1:
        return {"world": 1, "ms_per_step": round(step1_ms, 1)}
    link = XGMI_LINK_GBS * XGMI_EFFICIENCY * 1e9
    total = seg_bytes * n_layers + tail_bytes
    out = {"world": world, "assumptions": {"xgmi_link_GBps": XGMI_LINK_GBS, "link_efficiency": XGMI_EFFICIENCY,
                                           "gemm_slowdown_while_collective_resident": GEMM_SLOWDOWN_UNDER_COLLECTIVE,
                                           "one_rank_step_ms": round(step1_ms, 1), "one_rank_adamw_ms": round(adamw_ms, 1),
                                           "grad_bytes_per_layer_segment": int(seg_bytes), "grad_bytes_outside_decoder": int(tail_bytes)}}
    layer_bwd_ms = layer_bwd_ms if layer_bwd_ms is not None else (step1_ms - adamw_ms) * (2.0 / 3.0) / n_layers
    for algo, per_link in (("ring", (world - 1) / world), ("direct", 1.0 / world)):
        rs_seg = seg_bytes * per_link / link * 1e3                       # ms one layer's reduce-scatter keeps the links (and some CUs) busy
        rs_tail = tail_bytes * per_link / link * 1e3                     # exposed: issued at step(), nothing left to hide it
        ag_all = total * per_link / link * 1e3                           # exposed with the synchronous update (async_update hides most of it)
        hidden = min(rs_seg, layer_bwd_ms)                               # a segment's reduce-scatter hides under the next layer's backward
        exposed_rs = rs_tail + n_layers * (rs_seg - hidden) + rs_seg     # + the last layer's segment (nothing behind it)
        slow = GEMM_SLOWDOWN_UNDER_COLLECTIVE * gemm_share * (n_layers * hidden)
        compute = step1_ms - adamw_ms * (1.0 - 1.0 / world)
        ms = compute + exposed_rs + ag_all + slow
        out[algo] = {"ms_per_step": round(ms, 1), "scaling_vs_one_rank": round(world * step1_ms / ms, 2),
                     "exposed_reduce_scatter_ms": round(exposed_rs, 1), "all_gather_ms": round(ag_all, 1), "gemm_slowdown_ms": round(slow, 1),
                     "compute_ms": round(compute, 1)}
    return out


def cpu_baseline(args):
    """The CPU oracle (oracle/ref_model.py, kind 'port') on a bounded sample of the same workload: ONE sample built like the bench's
    (one 256-token image + text, labels as in make_batch) at the workload's own length (2048 spliced tokens), LLaMA-3-8B / SO400M layer
    geometry with FOUR of 32 decoder layers and TWO of 27 tower layers actually run plus the full 128258-entry lm_head and both loss
    heads, fp32, forward+backward with the stage-2 freeze policy.  Every stage is timed on its own -- each decoder layer's forward
    separately (so the per-layer cost is a mean of four measurements, not a difference of two), the heads' forward, the heads' backward and the decoder's
    backward as two separate autograd passes -- and the per-layer figures are scaled to the full depth to quote tokens/s."""
    from oracle.ref_model import OracleConfig, init_state_dict
    from oracle import ref_model as RM, ref_ops as R
    threads = _cpu_threads()
    torch.set_num_threads(threads)
    NL = 4                                           # decoder layers actually run (per-layer cost = mean of four separately timed layers)
    NV = 2                                           # tower layers actually run (frozen: forward only, scaled to 27)
    cfg = OracleConfig(num_hidden_layers=NL, v_layers=NV, num_image_tokens=args.image_tokens, tokenizer_model_max_length=4096)
    sd = init_state_dict(cfg, seed=1, fast_big=True)
    for k, v in sd.items():
        if "vision_tower" not in k and "vision_proj" not in k:
            v.requires_grad_(True)
    L = args.seq
    ids, labels, mask, images = make_batch(1, L, args.image_tokens, "cpu", seed=0, frames=1, all_generation=True)
    images = images.float()
    t0 = time.time()
    with torch.no_grad():
        feat = RM.vision_features(sd, cfg, images)
    t_vit = time.time() - t0
    t0 = time.time()
    proj = RM.mm_projector(sd, cfg, feat)
    x, lab, valid, pos, tgt, _ = RM.splice(sd, cfg, ids, labels, mask, proj, feat)
    t_splice = time.time() - t0
    cos, sin = R.rope_tables(torch.arange(L)[None], cfg.head_dim, cfg.rope_theta, x.dtype)
    t_layers = []
    x_leaf = x.detach().requires_grad_(True)         # the embedding / projector backward is a per-step constant: timed on its own below
    h = x_leaf
    for i in range(NL):                              # each layer's forward on its own clock
        t0 = time.time()
        h = RM.llama_layer(sd, cfg, i, h, valid, cos, sin)
        t_layers.append(time.time() - t0)
    t0 = time.time()
    hid_leaf = R.rmsnorm(h, sd["model.norm.weight"], cfg.rms_norm_eps).detach().requires_grad_(True)
    res = RM.heads(sd, cfg, hid_leaf, lab, pos, tgt, return_logits=False, ce_rows_only=True)
    t_head_f = time.time() - t0
    t0 = time.time()
    res["loss"].backward()                           # lm_head / vision_head gradients + d loss / d hidden
    t_head_b = time.time() - t0
    t0 = time.time()
    R.rmsnorm(h, sd["model.norm.weight"], cfg.rms_norm_eps).backward(hid_leaf.grad)      # final norm + NL decoder layers
    t_dec_b = time.time() - t0
    t0 = time.time()
    x.backward(x_leaf.grad)                          # splice: dense [V, h] embedding gradient (as torch's nn.Embedding produces) + projector
    t_emb_b = time.time() - t0
    n_ce = int((lab[:, 1:] != -100).sum())
    per_layer = (sum(t_layers) + t_dec_b) / NL
    t_head = t_head_f + t_head_b + t_splice + t_emb_b
    full = per_layer * 32 + t_head + t_vit * (27 / NV)
    return {"value": round(L / full, 3), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": (f"oracle/ref_model.py fp32 fwd+bwd, 1 sample of {L} spliced tokens ({args.image_tokens} image + {L - args.image_tokens} text; "
                       f"{n_ce} CE rows, {args.image_tokens} regression rows), LLaMA-3-8B + SO400M layer geometry with {NL}/32 decoder and "
                       f"{NV}/27 tower layers + full lm_head; measured per stage: decoder layer forward " + " / ".join(f"{t:.2f}s" for t in t_layers)
                       + f", decoder backward ({NL} layers) {t_dec_b:.2f}s, heads fwd {t_head_f:.2f}s bwd {t_head_b:.2f}s, embedding + projector bwd {t_emb_b:.2f}s, tower ({NV} layers) {t_vit:.2f}s; "
                       f"per-layer cost {per_layer:.2f}s (mean of {NL} separately timed layers) x 32 + heads + tower x 27/{NV} = {full:.1f}s per {L} tokens"),
            # two explicit fields (not prose): what the host clock measured for the layers actually run, and what `value` is quoted on
            "measured_seconds": round(sum(t_layers) + t_dec_b + t_head + t_vit, 2),
            "extrapolated_seconds_full_depth": round(full, 2),
            "layers_run": {"decoder": NL, "decoder_full": 32, "tower": NV, "tower_full": 27},
            "decoder_layer_forward_seconds": [round(t, 3) for t in t_layers]}


def cpu_baseline_c1(budget_s=60.0):
    """BASELINE configs[0] / BASELINE.md section 3 as planned: TinyLlama-1.1B geometry (22 layers, h 2048, 32/4 heads, I 5632, V 32002)
    + SigLIP-SO400M/14-384 (27 layers), 1 prompt image -> 256 tokens + 128 text ids (spliced L = 383), B = 1, stage-1 freeze policy
    (only mm_projector + embed_tokens train, reference train.py:1515-1519), the FULL model, directly timed forward+backward on the host
    cores: fp32 and bf16, each 1 warm-up + 3 timed steps (bf16 skipped when a probe GEMM shows the host has no fast bf16 path)."""
    from oracle.ref_model import OracleConfig, forward, init_state_dict
    threads = _cpu_threads()
    torch.set_num_threads(threads)
    cfg = OracleConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32, num_key_value_heads=4,
                       vocab_size=32002, rope_theta=10000.0, num_image_tokens=256, tokenizer_model_max_length=2048, image_start_id=32000,
                       use_vision_ar=False)
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(3, 31999, (1, 129), generator=g)
    ids[0, 0] = 1
    ids[0, 21], ids[0, 22], ids[0, 23] = 32000, -200, 32001
    labels = torch.full_like(ids, -100)
    labels[0, -64:] = ids[0, -64:]
    mask = torch.ones_like(ids, dtype=torch.bool)
    images = torch.randn(1, 3, 384, 384, generator=g)
    out = {"workload": "BASELINE configs[0]: TinyLlama-1.1B + SigLIP-SO400M/14-384 geometry, 1 image (256 tok) + 128 text, B=1, L=383, "
                       "stage-1 freeze (mm_projector + embed_tokens trainable), oracle/ref_model.py forward+backward, directly timed",
           "cores": threads, "unit": "tokens/s", "kind": "port"}
    t_all = time.time()
    sd32 = init_state_dict(cfg, seed=2, fast_big=True)
    for dt, tag, warm, n in ((torch.float32, "fp32", 1, 3), (torch.bfloat16, "bf16", 1, 3)):
        if tag == "bf16":                                  # probe: one bf16 GEMM of the MLP shape; hosts without fast bf16 GEMMs skip
            a, b = torch.randn(383, 2048).to(dt), torch.randn(5632, 2048).to(dt)
            torch.nn.functional.linear(a, b)
            t0 = time.time()
            torch.nn.functional.linear(a, b)
            probe = time.time() - t0
            fp32_probe = out["fp32"]["step_seconds"]
            if probe * 22 * 3 * 3 * 4 > 3 * fp32_probe or time.time() - t_all > budget_s:
                out["bf16"] = {"skipped": f"bf16 GEMM probe {probe * 1e3:.0f} ms (383x2048x5632): a bf16 step would take several times the fp32 "
                                          f"step ({fp32_probe:.1f}s) on this host / time budget {budget_s:.0f}s"}
                break
        sd = {k: v.to(dt) for k, v in sd32.items()} if dt != torch.float32 else sd32
        for k, v in sd.items():
            v.requires_grad_("mm_projector" in k or "embed_tokens" in k)
        ts = []
        for i in range(warm + n):
            for v in sd.values():
                v.grad = None
            t0 = time.time()
            r = forward(sd, cfg, ids, mask, labels, images.to(dt), return_logits=False, ce_rows_only=True)
            r["loss"].backward()
            if i >= warm:
                ts.append(time.time() - t0)
        step = sum(ts) / len(ts)
        out[tag] = {"step_seconds": round(step, 3), "value": round(383 / step, 2), "steps_timed": n, "warmup_steps": warm,
                    "step_seconds_each": [round(t, 3) for t in ts], "loss": round(float(r["loss"].detach()), 4)}
    return out


def rccl_summary(path, opt, steps, world):
    """What RCCL says it did (NCCL_DEBUG=INFO, rank 0's log): version, channels, per-collective algorithm / protocol lines, plus the
    achieved bus bandwidth of the gradient reduce-scatters and parameter all-gathers from the optimizer's own HIP-event timing."""
    import re
    out = {"log": path}
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return out
    m = re.search(r"RCCL version\s*:?\s*([^\n]+)", text) or re.search(r"NCCL version\s*:?\s*([^\n]+)", text)
    if m:
        out["version"] = m.group(1).strip()[:120]
    ch = re.findall(r"(\d+) coll channels", text) or re.findall(r"[Cc]hannels?[ =:]+(\d+)", text)
    if ch:
        out["coll_channels"] = int(ch[-1])
    # whatever the parse finds, the record carries the head of RCCL's own log (INIT lines: version, topology, channels, transports)
    out["log_lines"] = len(text.splitlines())
    keep = re.compile(r"RCCL version|coll channels|nranks|Channel \d+[/ ]|[Rr]ing|[Tt]ree|via |Algo|algo|proto|XGMI|xgmi|P2P|nchannels|nChannels")
    out["log_head"] = [ln.strip()[:160] for ln in text.splitlines() if "NCCL INFO" in ln and keep.search(ln)][:20]
    algo = {}
    for line in text.splitlines():                           # TUNING / COLL lines: "ReduceScatter: ... Algo RING proto SIMPLE ... nchannels N"
        m = re.search(r"(AllReduce|ReduceScatter|AllGather)[^\n]*?[Aa]lgo(?:rithm)? (\w+)[^\n]*?proto(?:col)? (\w+)(?:[^\n]*?(?:channels|nchannels|nChannels)[ =:{]*(\d+))?", line)
        if m:
            key = f"{m.group(1)}: algo {m.group(2)} proto {m.group(3)}" + (f" channels {m.group(4)}" if m.group(4) else "")
            algo[key] = algo.get(key, 0) + 1
    if algo:
        out["collectives_seen"] = dict(sorted(algo.items(), key=lambda kv: -kv[1])[:8])
    xgmi = len(re.findall(r"via P2P|XGMI|via SHM|via NET", text))
    if xgmi:
        out["transport_lines"] = {"p2p_or_xgmi": len(re.findall(r"via P2P|XGMI", text)), "shm": len(re.findall(r"via SHM", text)), "net": len(re.findall(r"via NET", text))}
    if hasattr(opt, "comm_bytes_per_step"):
        out["bytes_per_step"] = opt.comm_bytes_per_step()
    return out


def bus_bandwidth(rec_rccl, comm, world, overlap):
    """Bus bandwidth (bytes * (world - 1) / world / time) of the collectives whose WHOLE duration sits in an exposed wait: the parameter
    all-gather with the synchronous update, and the gradient reduce-scatter when the overlap is off (with the overlap on only its
    un-hidden remainder is timed, so no rate can be quoted)."""
    by = rec_rccl.get("bytes_per_step") if rec_rccl else None
    if not by or not comm or world < 2:
        return None
    f = (world - 1) / world
    out = {}
    if comm.get("all_gather"):
        out["all_gather_gb_s"] = round(by["all_gather_bytes"] * f / (comm["all_gather"] * 1e-3) / 1e9, 1)
    if not overlap and comm.get("reduce_scatter_exposed"):
        out["reduce_scatter_gb_s"] = round(by["reduce_scatter_bytes"] * f / (comm["reduce_scatter_exposed"] * 1e-3) / 1e9, 1)
    return out or None


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc:
        raise SystemExit(rc)


def newest_profiles(stem):
    """profiles/r<N>_<stem>*.json, newest round first (the counter evidence the bench line cites: taken on the shipped kernels of that round)."""
    import glob
    import re
    found = []
    for pth in glob.glob(os.path.join(REPO, "profiles", f"r*_{stem}*.json")):
        m = re.match(r"r(\d+)_", os.path.basename(pth))
        if m:
            found.append((int(m.group(1)), os.path.basename(pth), pth))
    return [pth for _, _, pth in sorted(found, reverse=True)]


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if (args.gpus > 1 or os.environ.get("MM355_BENCH_SELF_LAUNCH") == "1") and "RANK" not in os.environ:   # (=1: exercise the launcher on one GPU)
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU over RCCL), exactly the command line the
        # docstring gives; rank 0 of the child job prints the ONE JSON line on this process's stdout
        return self_launch(args.gpus)
    torch.cuda.set_device(local)
    if args.alloc_roundup > 0:
        torch.cuda.memory._set_allocator_settings(f"roundup_power2_divisions:{args.alloc_roundup}")
    dev = torch.device("cuda", local)
    for kv in args.set_variant:
        import metamorph_amd.functional as F_
        name, _, val = kv.partition("=")
        F_.set_variant(name, int(val or 1))
    # MM355_BENCH_FORCE_DIST=1: drive the RCCL call pattern with a single rank under torchrun (collectives forced at world size 1)
    force_dist = os.environ.get("MM355_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
    if force_dist:
        from metamorph_amd.zero2 import set_collective_mode
        set_collective_mode(force_collectives=True)
    rccl_log = None
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's own account of what it chose (algorithm / protocol / channels per collective): NCCL_DEBUG=INFO into a per-rank file,
        # rank 0's is summarised into the JSON line (the driver's 8-GPU run cannot be observed otherwise)
        rccl_log = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mm355_bench_rccl_{os.getpid()}_{rank}.log")
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION", "WARN"):   # (the GPU boxes export NCCL_DEBUG=VERSION)
            os.environ["NCCL_DEBUG"] = "INFO"
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL,TUNING")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        # stdout carries exactly ONE line (the JSON record): RCCL prints its NCCL_DEBUG=VERSION banner with printf when the
        # communicator is created, so fd 1 points at stderr while that happens
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)      # backend "nccl" IS RCCL on ROCm
            dist.barrier()                                       # forces communicator creation now
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from metamorph_amd.zero2 import Zero2AdamW, tag_segments

    t_build = time.time()
    model = build_bench_model(dev, layers=args.layers, vit_layers=args.vit_layers, image_tokens=args.image_tokens, llama31_rope=args.llama31_rope)
    if args.train_vision:                                        # reference: freeze_vision=False + `vision_lr` parameter group
        tower = model.get_model().vision_tower
        tower.freeze_vision = False
        for n, p in tower.named_parameters():
            p.requires_grad_("post_layernorm" not in n)
    params = [p for p in model.parameters() if p.requires_grad]
    n_params = sum(p.numel() for p in params)
    if world > 1:                                                # identical initial weights on every rank
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    tag_segments(model)                                          # one gradient-reduction segment per decoder layer
    if args.train_vision:
        vis = {id(p) for p in model.get_model().vision_tower.parameters()}
        params = [dict(params=[p for p in params if id(p) not in vis], lr=2e-5), dict(params=[p for p in params if id(p) in vis], lr=2e-6)]
    ckpt_layers = None
    if args.grad_checkpointing:
        model.gradient_checkpointing_enable()
        if args.ckpt_layers != "all":
            nl_ = len(model.get_model().layers)
            if args.ckpt_layers == "auto":
                # what a decoder layer keeps for its backward when it is NOT recomputed: layer input, post-RoPE qkv, attention output,
                # post-attention residual, gate|up (bf16 rows) + lse / rstd (fp32): DESIGN.md section 5; fixed state = parameters,
                # bf16 gradients, fp32 Adam moments (+ the frozen tower) plus the step's transients (logits rows, backward buffers)
                cfg_ = model.config
                rows_ = args.batch * args.seq
                hd_ = cfg_.hidden_size // cfg_.num_attention_heads
                per_layer = rows_ * 2 * (3 * cfg_.hidden_size + (cfg_.num_attention_heads + 2 * cfg_.num_key_value_heads) * hd_
                                         + 2 * cfg_.intermediate_size) + rows_ * 4 * (cfg_.num_attention_heads + 2)
                fixed = 12 * n_params / max(world, 1) * (1 if args.zero else world) + 2 * n_params + 2.0e9
                transient = rows_ * 2 * (6 * cfg_.intermediate_size + 8 * cfg_.hidden_size) + 8.0e9
                total_mem = torch.cuda.get_device_properties(dev).total_memory
                room = 0.90 * total_mem - fixed - transient - nl_ * rows_ * 2 * cfg_.hidden_size
                keep = max(0, min(nl_, int(room // (per_layer - rows_ * 2 * cfg_.hidden_size))))
                ckpt_layers = nl_ - keep
            else:
                ckpt_layers = max(0, min(nl_, int(args.ckpt_layers)))
            model.get_model().checkpoint_layers = ckpt_layers
    if args.zero == 3:
        from metamorph_amd.zero3 import Zero3AdamW
        opt = Zero3AdamW(params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0).enable_hooks()
    else:
        opt = Zero2AdamW(params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                         async_update=bool(args.zero2_async))
        if args.zero2_overlap:
            opt.enable_overlap()
    t_build = time.time() - t_build

    # a pool of distinct seeded batches, all resident before the timed region, rotated through the steps
    n_pool = max(1, args.pool)
    pool = [make_batch(args.batch, args.seq, args.image_tokens, None if args.host_inputs else dev, seed=1234 + rank + 1000 * j,
                       frames=args.frames, all_generation=args.all_generation, ragged=args.ragged) for j in range(n_pool)]
    model.config.mm355_compact_rows = {"auto": "auto", "on": True, "off": False, "exact": "exact"}[args.compact_rows]
    # valid spliced rows of every pool batch (mask rows + the image rows the splice inserts): what `value` counts under --ragged
    pool_valid = [int(b_[2].sum()) + args.batch * args.frames * (args.image_tokens - 1) for b_ in pool]
    valid_timed = [0]
    host_bytes = sum(t.numel() * t.element_size() for t in pool[0]) if args.host_inputs else 0
    timer = GemmTimer()
    hbm_timer = HbmTimer()
    if not args.no_kernel_timing:
        timer.install()
        hbm_timer.install()
    step_no = [0]
    from metamorph_amd import hostmirror

    def step():
        ids, labels, mask, images = pool[step_no[0] % n_pool]
        valid_timed[0] += pool_valid[step_no[0] % n_pool]
        step_no[0] += 1
        if args.host_inputs:                                 # host -> HBM inside the step (HF Trainer._prepare_inputs), pixels cast on the device
            ids, labels, mask = (hostmirror.to_device(t, dev, non_blocking=True) for t in (ids, labels, mask))
            images = images.to(dev, non_blocking=True).to(torch.bfloat16)
        opt.zero_grad()
        out = model(input_ids=ids, attention_mask=mask, labels=labels, images=images)
        if args.zero2_overlap:
            opt.arm_overlap()                                    # no accumulation: these gradients are final
        out.loss.backward()
        opt.step()
        return out.loss

    loss_first = None                                            # loss of the very first step: the model as built, before any update
    for _ in range(args.warmup):
        loss = step()
        loss_first = loss.detach() if loss_first is None else loss_first
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = hbm_timer.enabled = True
    valid_timed[0] = 0
    dist_run = world > 1 or force_dist
    if dist_run and hasattr(opt, "comm_timing"):
        opt.comm_timing = True                                   # HIP events around the parts of step() that wait for RCCL
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
        loss_first = loss.detach() if loss_first is None else loss_first
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = hbm_timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss_val = float(loss.detach())
    comm = opt.comm_summary(args.steps) if (dist_run and hasattr(opt, "comm_summary")) else None
    if comm is not None:
        opt.comm_timing = False
    # ---- multi-rank hygiene, outside the timed region: every rank must hold the same parameters after the same updates
    params_equal = None
    ab_async = None
    if dist_run:
        if hasattr(opt, "synchronize"):
            opt.synchronize()
        chk = torch.zeros(2, device=dev, dtype=torch.float64)
        for p in model.parameters():
            if p.numel():
                f = p.data.view(-1)
                chk[0] += f.double().sum()
                chk[1] += (f[:: max(1, f.numel() // 4096)].double() ** 2).sum()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        params_equal = bool(torch.equal(lo, hi))
        # same-job A/B of the asynchronous update / all-gather (hidden behind the next forward pass) against the synchronous one:
        # 3 steps each AFTER the headline measurement (the timed region and the record above are closed; default at world > 1, where it
        # is the only way to see the overlap on a node this build never runs on; MM355_BENCH_AB_ASYNC=1 forces it on one rank)
        if args.zero == 2 and (world > 1 or os.environ.get("MM355_BENCH_AB_ASYNC", "0") == "1"):
            ab_async = {}
            for tag, on in (("sync_ms_per_step", False), ("async_ms_per_step", True)):
                opt.set_async_update(on)
                step()
                torch.cuda.synchronize()
                dist.barrier()
                ta = time.perf_counter()
                for _ in range(3):
                    step()
                opt.synchronize()
                torch.cuda.synchronize()
                dist.barrier()
                tt = torch.tensor([(time.perf_counter() - ta) / 3 * 1e3], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ab_async[tag] = round(float(tt), 2)
            opt.set_async_update(False)
    tokens_per_rank = args.batch * args.seq
    value = world * tokens_per_rank * args.steps / dt
    if args.ragged > 0:                                          # valid tokens only (this rank's count x ranks: every rank draws the same distribution)
        value = world * valid_timed[0] / dt

    n_all, t_all_g, fl_all = timer.summary() if not args.no_kernel_timing else (0, 0.0, 0.0)
    n_gemm, t_gemm, fl_gemm, bytes_gemm = timer.summary(plain_only=True) if not args.no_kernel_timing else (0, 0.0, 0.0, 0.0)
    n_fused, t_fused, fl_fused = timer.fused_summary() if not args.no_kernel_timing else (0, 0.0, 0.0)
    roofline = None
    if n_all:
        # The dominant kernel = the bf16 MFMA GEMM family: EVERY GEMM-bearing launch of the timed steps (plain ping-pong launches, their
        # two-problem and RoPE-epilogue forms, the small-tile kernel, and the two fused MLP launches, whose GEMM flops are taken over their
        # whole duration, fused element-wise epilogue included).  `plain` / `fused_mlp` split the same launches.
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE with the gfx950 x2 read
        # correction) over this same command, summarised by tools/hbm_traffic_summary.py and committed under profiles/; they only
        # apply to the configuration they were taken on.  Mean over all GEMM launches, weighted by launch count.
        traffic, traffic_src = None, None
        canonical = args.batch == 16 and args.layers == 32 and args.seq == 2048 and args.frames == 1
        for tpath in newest_profiles("step_b16_hbm_traffic") if canonical else ():      # newest round first
            ks = [k for k in json.load(open(tpath))["kernels"] if k["kernel"].startswith("gemm_")]
            if ks:
                traffic = round(sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / sum(k["launches"] for k in ks))
                traffic_src = "profiles/" + os.path.basename(tpath)
                break
        # MFMA-pipe utilisation from the counters (north_star: "evidenced by rocprof MFMA utilisation"): SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs /
        # active cycles, one rocprofv3 --pmc pass over this same command (tools/pmc_mfma_busy.py); GEMM family weighted by active cycles
        mfma_busy, mfma_busy_src, mfma_busy_attn = None, None, None
        for bpath in newest_profiles("step_b16_mfma_busy") if canonical else ():
            ks = json.load(open(bpath))["kernels"]
            gs = [k for k in ks if k["kernel"].startswith("gemm_")]
            if gs:
                wsum = sum(k["active_cycles_per_launch"] * k["launches"] for k in gs)
                mfma_busy = round(sum(k["mfma_busy"] * k["active_cycles_per_launch"] * k["launches"] for k in gs) / wsum, 4)
                mfma_busy_attn = {k["kernel"]: round(k["mfma_busy"], 4) for k in ks if k["kernel"].startswith("attn4")}
                mfma_busy_src = "profiles/" + os.path.basename(bpath)
                break
        ach = fl_all / t_all_g / 1e12
        bytes_all = sum(r[4] for r in timer.records)
        roofline = {"bound": "mfma", "kernel": "bf16 MFMA GEMM family, ALL launches of the timed steps: gemm_pp_kernel, gemm_pp_pair_kernel (two problems, one grid), "
                                               "gemm_pp_rope_kernel (RoPE in the q|k|v epilogue), gemm_nt_kernel (small tiles), gemm_pp_swiglu_kernel and "
                                               "gemm_pp_swiglu_bwd_kernel (SwiGLU forward / backward in the MLP GEMMs' epilogues; GEMM flops over the whole launch)",
                    "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (L2 misses incl. Infinity-Cache hits), mean over all GEMM launches",
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(bytes_all / n_all),
                    "mfma_busy": mfma_busy, "mfma_busy_attention": mfma_busy_attn, "mfma_busy_source": mfma_busy_src,
                    "mfma_busy_unit": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8): share of the dispatch's active cycles the matrix pipe is busy",
                    "launches": n_all, "gemm_seconds_per_step": round(t_all_g / args.steps, 4),
                    "algorithmic_flops_per_step": fl_all / args.steps}
        if n_gemm:
            roofline["plain"] = {"kernel": "gemm_pp_kernel + gemm_pp_pair_kernel + gemm_pp_rope_kernel + gemm_nt_kernel", "launches": n_gemm,
                                 "achieved": round(fl_gemm / t_gemm / 1e12, 1), "frac": round(fl_gemm / t_gemm / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                 "seconds_per_step": round(t_gemm / args.steps, 4), "algorithmic_flops_per_step": fl_gemm / args.steps,
                                 "algorithmic_bytes_per_launch": round(bytes_gemm / n_gemm)}
        if n_fused:
            # gate|up GEMM + SwiGLU and down_proj input-gradient GEMM + SwiGLU backward: the epilogue also carries the element-wise pass it
            # absorbed (HBM-bound there), so this is a floor for the MFMA part, not a like-for-like figure
            roofline["fused_mlp"] = {"kernel": "gemm_pp_swiglu_kernel + gemm_pp_swiglu_bwd_kernel", "launches": n_fused,
                                     "achieved": round(fl_fused / t_fused / 1e12, 1), "frac": round(fl_fused / t_fused / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                     "seconds_per_step": round(t_fused / args.steps, 4), "algorithmic_flops_per_step": fl_fused / args.steps}
        roofline["hbm_bound"] = hbm_timer.summary(args.steps)   # RMSNorm / transposes / AdamW: algorithmic bytes over launch time vs 8 TB/s
    # whole-step model flops: 3 x (32 x (436.2 MFLOP + 2 L h) + 2 h V) per token + 666.5 GFLOP per image
    h, V, L = 4096, 128258, args.seq
    per_tok = 3.0 * (args.layers * (436.2076e6 + 2.0 * L * h) + 2.0 * h * V)
    flop_tokens = valid_timed[0] / max(args.steps, 1) if args.ragged > 0 else tokens_per_rank       # model flops are those of the VALID tokens
    step_flops = per_tok * flop_tokens + args.batch * args.frames * 666.5e9 * (args.vit_layers / 27.0) * (3.0 if args.train_vision else 1.0)
    mfu = step_flops * args.steps / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS

    if rank == 0:
        rec = {
            "metric": "train tokens/sec (LLaMA-3-8B + SigLIP-SO400M, seq2048, 256 img toks), whole job",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random-init weights, random tokens/pixels)" + (
                f"; INPUTS FROM PINNED HOST MEMORY every step ({host_bytes} B/step over PCIe: not the headline configuration)" if args.host_inputs else ""),
            "tokens_per_sec_per_gpu": round(value / world, 1),
            "config": {"workload": "BASELINE configs[1]: LLaMA-3-8B + SigLIP-SO400M/14-384, spliced seq 2048 with one 256-token image per sample, "
                                   "bf16 full fine-tune (tower " + ("trainable" if args.train_vision else "frozen") + "), AdamW + ZeRO-2" + (
                                       "; LLaMA-3.1 rope_scaling (rope_type llama3) instead of BASELINE's plain RoPE" if args.llama31_rope else "") + (
                                       f"; RAGGED batches, mean padding share {args.ragged:.2f} (value = VALID tokens/s: {valid_timed[0] / max(args.steps, 1) / (args.batch * args.seq):.3f} "
                                       f"of the padded rows; decoder rows {getattr(model, '_decoder_rows', None)}; compact rows {args.compact_rows}) -- NOT the headline" if args.ragged > 0 else ""),
                       "global_batch": world * args.batch, "per_gpu_batch": args.batch, "seq_len": args.seq, "image_tokens": args.image_tokens, "frames_per_sample": args.frames,
                       "decoder_layers": args.layers, "tower_layers": args.vit_layers, "trainable_params": n_params,
                       **({"variants": sorted(args.set_variant)} if args.set_variant else {}),
                       "parallelism": f"dp{world} zero{args.zero}" + ((" +recompute" + ("" if ckpt_layers is None else f"(first {ckpt_layers} layers)")) if args.grad_checkpointing else "") + (
                           " +async-update" if getattr(opt, "async_update", False) else ""), "samples": (f"{args.batch} image-generation per GPU" if args.all_generation else f"{args.batch - 1} image-QA + 1 image-generation per GPU")},
            "loss": round(loss_val, 4), "loss_step0": round(float(loss_first), 5), "batch_pool": n_pool, "model_tflops_per_gpu": round(step_flops * args.steps / dt / 1e12, 1),
            "mfu_vs_bf16_mfma_peak": round(mfu, 4), "build_seconds": round(t_build, 1),
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1),
        }
        if roofline:
            rec["roofline"] = roofline
        # the pre-registered multi-GPU expectation: at one rank for 2 / 4 / 8 ranks from THIS run's step time, at N > 1
        # for this N from the committed one-rank reference -- beside the measured `ms_per_step`
        seg_b, tail_b = 218_112_000 * 2.0, (n_params - 32 * 218_112_000) * 2.0 if args.layers == 32 else 0.0
        if args.layers == 32 and not args.train_vision and args.zero == 2:
            adamw_ms = 39.4 * (n_params / 8.07e9)
            if world == 1:
                rec["predicted_scaling"] = [predict_step(n, dt / args.steps * 1e3, adamw_ms, seg_b, tail_b) for n in (2, 4, 8)]
            else:
                ref_ms = ONE_RANK_REFERENCE_MS.get((args.batch, args.seq, args.frames))
                if ref_ms:
                    rec["predicted"] = predict_step(world, ref_ms, adamw_ms, seg_b, tail_b)
                    rec["predicted_ms_per_step"] = {k: rec["predicted"][k]["ms_per_step"] for k in ("ring", "direct")}
        rec["rccl_ranks"] = dist.get_world_size() if (world > 1 or force_dist) else 1
        if dist_run:
            rec["params_equal_across_ranks"] = params_equal
            rec["comm_ms_per_step"] = comm                       # exposed (not overlapped) time per phase, rank 0's stream
            rec["zero2_overlap"] = bool(args.zero2_overlap)
            rec["rccl"] = rccl_summary(rccl_log, opt, args.steps, world)
            rec["rccl"]["bus_bandwidth"] = bus_bandwidth(rec["rccl"], comm, world, bool(args.zero2_overlap) and getattr(opt, "overlap", True))
            if ab_async:
                rec["zero2_async_ab"] = ab_async
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args)
            rec["cpu_baseline_c1"] = cpu_baseline_c1()
        print(json.dumps(rec), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if dist_run and params_equal is False:                       # ranks diverged: the number above is not a training run
        print("bench.py: parameters differ across ranks after the timed steps (params_equal_across_ranks = false)", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
