#!/usr/bin/env python3
"""Decode throughput of the cached greedy loop (row N1) at LLaMA-3-8B geometry: ms/token and the weight-streaming rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamorph_amd.factory import LLAMA3_8B, build_model

dev = torch.device("cuda:0")
layers = int(os.environ.get("LAYERS", 32))
model = build_model(dict(LLAMA3_8B, num_hidden_layers=layers), dict(num_hidden_layers=1), num_image_tokens=256, max_length=4096,
                    device=dev, init_on_device=True).eval()
h = 4096
L0, new = int(os.environ.get("PROMPT", 512)), int(os.environ.get("NEW", 64))
emb = (torch.randn(1, L0, h, device=dev) * 0.02).bfloat16()
for use_cache in (() if os.environ.get("NO_GREEDY") else (True,) if os.environ.get("CACHED_ONLY") else (True, False)):
    n = new if use_cache else min(new, 8)
    model.greedy_decode(None, None, emb, max_new_tokens=2, use_cache=use_cache, eos_token_id=())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.greedy_decode(None, None, emb, max_new_tokens=n, use_cache=use_cache, eos_token_id=())[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    wbytes = sum(p.numel() for n_, p in model.named_parameters() if "vision_tower" not in n_ and "embed_tokens" not in n_) * 2
    print(f"use_cache={use_cache}: {out.numel()} tokens in {dt*1e3:.1f} ms = {dt/out.numel()*1e3:.2f} ms/token"
          f" (prompt {L0}; weights {wbytes/1e9:.1f} GB -> {wbytes*out.numel()/dt/1e12:.2f} TB/s if streamed once per token)", flush=True)
    if use_cache:                                             # the per-token step alone: the difference of two run lengths (prompt pass cancels)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out2 = model.greedy_decode(None, None, emb, max_new_tokens=2 * n, use_cache=True, eos_token_id=())[0]
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
        per = (dt2 - dt) / max(out2.numel() - out.numel(), 1)
        print(f"   decode step alone: {per*1e3:.3f} ms/token = {wbytes/per/1e12:.2f} TB/s of weights; prompt pass + first token ~ {(dt - per*out.numel())*1e3:.1f} ms", flush=True)

# ---- the prompt pass alone (round 6): decoder_prefill over L0 rows through the training-path kernels, KV rows kept; floor = the later of
#      streaming the weights once (15.1 GB at the ~6.3 TB/s achievable) and the GEMM + attention flops at the 1.45 PF the big GEMMs sustain
if not os.environ.get("NO_PREFILL"):
    import metamorph_amd.functional as F
    wbytes = sum(p.numel() for n_, p in model.named_parameters() if "vision_tower" not in n_ and "embed_tokens" not in n_ and "lm_head" not in n_) * 2
    with torch.no_grad():
        for Lp in [int(x) for x in os.environ.get("PROMPTS", "128,512,2048").split(",")]:
            _, meta = model._decode_meta(Lp)
            cos, sin = model.model.rope_tables(Lp + 8, dev)
            meta.cos, meta.sin = cos, sin
            kv = F.KVCache(len(model.model.layers), Lp + 8, meta.Hkv * meta.d, dev, Hq=meta.Hq, d=meta.d)
            x = (torch.randn(Lp, h, device=dev) * 0.02).bfloat16()
            for _ in range(2):
                kv.set_lengths([0]); F.decoder_prefill(x, model.model.layers, meta, kv)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                kv.set_lengths([0]); F.decoder_prefill(x, model.model.layers, meta, kv)
            torch.cuda.synchronize(); per = (time.perf_counter() - t0) / reps
            flops = layers * (436.2076e6 + 2.0 * Lp * h) * Lp
            floor = max(wbytes / 6.3e12, flops / 1.45e15)
            print(f"prompt pass {Lp:5d} rows: {per*1e3:7.2f} ms  ({flops/per/1e12:6.0f} TFLOP/s, weights at {wbytes/per/1e12:.2f} TB/s; floor {floor*1e3:.2f} ms = "
                  f"{'weights' if wbytes / 6.3e12 > flops / 1.45e15 else 'flops'}; x {per/floor:.2f})", flush=True)
            del kv

# ---- the batched cached step (round 5: all rows of a batch / all beams in ONE pass over the weights): ms per step by batch size
if not os.environ.get("NO_BATCH"):
    import metamorph_amd.functional as F
    if os.environ.get("FOLD_ROWS"):
        F.set_variant("decode_fold_rows", int(os.environ["FOLD_ROWS"]))
    wbytes = sum(p.numel() for n_, p in model.named_parameters() if "vision_tower" not in n_ and "embed_tokens" not in n_) * 2
    with torch.no_grad():
        for B in [int(b) for b in os.environ.get("BATCHES", "1,2,3,4,8,16,32").split(",")]:
            _, meta = model._decode_meta(L0)
            cap = L0 + 2 * new + 4
            cos, sin = model.model.rope_tables(cap, dev)
            meta.cos, meta.sin = cos, sin
            kv = F.KVCache(len(model.model.layers), cap, meta.Hkv * meta.d, dev, Hq=meta.Hq, d=meta.d, batch=B)
            kv.k.normal_(0, 0.5); kv.v.normal_(0, 0.5)
            kv.set_lengths([L0 - 7 * b for b in range(B)])                   # ragged lengths, as a left-padded batch has them
            st = F.DecodeStepGraph(model.model.layers, meta, kv, cos, sin, h, dev)
            rows = (torch.randn(B, h, device=dev) * 0.02).bfloat16()
            for _ in range(3):
                st.step(rows)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(new):
                st.step(rows)
            torch.cuda.synchronize(); per = (time.perf_counter() - t0) / new
            print(f"batched step B={B:2d}: {per*1e3:.3f} ms/step = {per/B*1e3:.3f} ms per token and sequence; weights at {wbytes/per/1e12:.2f} TB/s "
                  f"({'graph' if st.graph is not None else 'eager'})", flush=True)
            del st, kv
