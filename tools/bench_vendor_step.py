#!/usr/bin/env python3
"""Same-node VENDOR-STACK yardstick (tools only; never imported by the package or by bench.py).

What would the reference's own stack do on this MI355X?  The reference delegates every flop to stock PyTorch ops (hipBLASLt / rocBLAS
GEMMs, `scaled_dot_product_attention`, elementwise ATen kernels) through transformers' LlamaModel / SiglipVisionModel
(reference metamorph_llama.py:349-359, siglip_encoder.py:138-213).  This tool runs the repo's restatement of exactly that arithmetic --
`oracle/ref_model.forward` (test infrastructure: tower -> projector -> splice -> 32 decoder layers -> lm_head + CE -> image-AR head) -- in
bf16 ON THE GPU through those stock ops, with torch autograd for the backward pass and torch.optim.AdamW(fused=True) for the update, on the
same synthetic workload bench.py times (BASELINE configs[1]: LLaMA-3-8B + SO400M, spliced seq 2048, one 256-token image per sample), same box.

Differences, all in the vendor stack's favour or neutral:
  * attention: the oracle's fp32 [L, L] score matrix is replaced by torch's fused SDPA (is_causal, GQA) -- what HF runs (`sdpa`);
  * optimizer state: torch AdamW keeps bf16 moments for bf16 parameters (8 B/param less traffic than the product's fp32 master + moments);
  * batch: as many samples as fit (autograd keeps every intermediate; no recompute), default 8; tokens/s is per token so sizes compare.
Prints ONE JSON line.  It is a yardstick, not a product path: nothing here is shipped, and no number from here enters bench.py's line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch
import torch.nn.functional as TF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--vit-layers", type=int, default=27)
    ap.add_argument("--device", default="cuda:0", help="cpu: dry run of the tool's plumbing (tiny --layers / --batch)")
    args = ap.parse_args()
    import bench
    from oracle import ref_model as RM
    from oracle import ref_ops as RO
    dev = torch.device(args.device)
    gpu = dev.type == "cuda"
    if gpu:
        torch.cuda.set_device(dev)
    backend = {"name": None}

    def sdpa_attention(q, k, v, key_valid=None, causal=True, scale=None):
        # the bench batches are full-length (no padding): plain causal (decoder) / full (tower) attention; GQA by head-group broadcast
        assert key_valid is None or bool(key_valid.all())
        backend["name"] = "torch.nn.functional.scaled_dot_product_attention"
        return TF.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale, enable_gqa=q.shape[1] != k.shape[1])

    RO.attention = sdpa_attention
    RM.ops.attention = sdpa_attention
    cfg = RM.OracleConfig(num_hidden_layers=args.layers, v_layers=args.vit_layers, tokenizer_model_max_length=4096)
    t0 = time.time()
    # the SAME random weights bench.py times (its model builder, device generator, seed 1234), taken over as a flat state dict: the
    # product model is only the initialiser here and is dropped before anything runs
    model = bench.build_bench_model(dev, layers=args.layers, vit_layers=args.vit_layers, image_tokens=256)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    del model
    for k, v in sd.items():
        v.requires_grad_("vision_tower" not in k and "vision_proj" not in k)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, fused=True)
    n_params = sum(p.numel() for p in params)
    pool = [bench.make_batch(args.batch, args.seq, 256, "cpu", 1234 + i) for i in range(2)]
    pool = [(i_.to(dev), l_.to(dev), m_.to(dev), x_.to(dev).bfloat16()) for i_, l_, m_, x_ in pool]
    build_s = time.time() - t0
    torch.set_default_device(dev)          # the oracle's factory calls (zeros / arange / index tensors) then land next to the weights

    def step(i):
        ids, labels, mask, images = pool[i % len(pool)]
        opt.zero_grad(set_to_none=True)
        out = RM.forward(sd, cfg, ids, mask, labels, images, return_logits=False, ce_rows_only=True)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return out["loss"].detach()

    for i in range(args.warmup):
        step(i)
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    sync()
    dt = (time.perf_counter() - t0) / args.steps
    tok = args.batch * args.seq
    h, V, L = cfg.hidden_size, cfg.vocab_size, args.seq
    step_flops = 3.0 * (args.layers * (436.2076e6 + 2.0 * L * h) + 2.0 * h * V) * tok + args.batch * 666.5e9 * (args.vit_layers / 27.0)
    print(json.dumps({"what": "vendor-stack yardstick: oracle/ref_model.forward in bf16 on the GPU through stock PyTorch-ROCm ops (hipBLASLt GEMMs, fused SDPA), "
                              "torch autograd backward, torch.optim.AdamW(fused=True); same synthetic workload as bench.py, same box",
                      "tokens_per_s": round(tok / dt, 1), "ms_per_step": round(dt * 1e3, 1), "batch": args.batch, "seq_len": args.seq,
                      "decoder_layers": args.layers, "tower_layers": args.vit_layers, "trainable_params": n_params, "loss": round(float(loss), 4),
                      "model_tflops": round(step_flops / dt / 1e12, 1), "mfu_vs_bf16_mfma_peak": round(step_flops / dt / 2.5e15, 4),
                      "attention": backend["name"], "torch": torch.__version__, "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1) if gpu else None,
                      "build_seconds": round(build_s, 1)}), flush=True)


if __name__ == "__main__":
    main()
