#!/usr/bin/env python3
"""Which ATen (non-mm355) kernels does one training step launch, and from which Python line?  (hygiene audit: VERDICT r1 item 9)
usage: python tools/aten_audit.py [--layers 4] [--batch 2]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--batch", type=int, default=2)
args = ap.parse_args()

from bench import make_batch
from metamorph_amd.factory import LLAMA3_8B, build_model
from metamorph_amd.zero2 import Zero2AdamW, tag_segments

dev = torch.device("cuda", 0)
model = build_model(dict(LLAMA3_8B, num_hidden_layers=args.layers), dict(num_hidden_layers=2), num_image_tokens=256, max_length=4096, device=dev,
                    init_on_device=True)
model.train()
tag_segments(model)
opt = Zero2AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-5, max_grad_norm=1.0).enable_overlap()
ids, labels, mask, images = make_batch(args.batch, 2048, 256, dev, seed=1)


def step():
    opt.zero_grad()
    out = model(input_ids=ids, attention_mask=mask, labels=labels, images=images)
    opt.arm_overlap()
    out.loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=8)
rows = []
for ev in ka:
    dt = getattr(ev, "self_device_time_total", 0.0)
    if dt <= 0 or not (ev.key.startswith("aten::") or "Memcpy" in ev.key or "Memset" in ev.key):
        continue
    where = "?"
    for fr in ev.stack or []:
        if "/metamorph_amd/" in fr or "bench.py" in fr:
            where = fr.split("/repo/")[-1]
            break
    rows.append((dt, ev.count, ev.key, where))
rows.sort(reverse=True)
for dt, n, name, where in rows[:60]:
    print(f"{dt / 1e3:8.3f} ms {n:5d}x {name:28s} {where}")
print(f"total ATen / memcpy device time in one step ({args.layers} layers, batch {args.batch}): {sum(r[0] for r in rows) / 1e3:.3f} ms")
