"""ZeRO-3 style parameter sharding for the decoder layers (BASELINE configs[4]: LLaMA-3-70B + SigLIP-SO400M on 8 x MI355X;
reference scripts/zero3.json:16-27, selected by scripts/slurm_finetune.sh:105).

`Zero2AdamW` keeps the complete bf16 parameters and a complete gradient buffer on every rank: 4 B/param replicated + 12 B/param
sharded.  At 70 B parameters that is 280 GB + 105 GB per GPU on an 8-GPU node -- more than the 288 GB of HBM.  `Zero3AdamW` shards
what is big and keeps what is small:

  * every DECODER LAYER segment (the `_mm_segment = ("layer", i)` runs of `tag_segments`; 1.71 GB of bf16 at LLaMA-3-70B) exists
    only as this rank's 1/world slice: bf16 parameter shard + bf16 gradient shard + fp32 master / Adam moments;
  * right before a layer is used -- `functional.params_ready(layer)` in the forward pass, the cached decode and again at the start of
    `DecoderLayerFn.backward` -- its parameters are ALL-GATHERED (RCCL, in place: each rank's slice is already at its offset) into
    one of `param_slots` rotating full-size buffers and the layer's `p.data` are re-pointed there (the q/k/v and gate/up blocks stay
    adjacent, so the fused-GEMM views keep working).  The gather of the NEXT layer in walk order is issued asynchronously at the same
    time, so it overlaps this layer's kernels; xGMI is point-to-point, hence few large collectives (one per layer and direction);
  * the backward kernels write the layer's weight gradients into one of `grad_slots` rotating full-size buffers; when the layer's
    backward has finished (`functional.set_layer_grad_hook`) the slot is REDUCE-SCATTERED (sum; the 1/world mean is folded into the
    update) and this rank's slice is added to its persistent gradient shard -- on every micro-step, there is no full gradient buffer
    to accumulate in;
  * everything outside the decoder layers (embeddings, final norm, projector, lm_head, vision head, a trainable tower: 4.2 GB at
    70 B) stays RESIDENT exactly as under ZeRO-2 (DeepSpeed keeps small tensors resident too: `stage3_param_persistence_threshold`);
  * step(): reduce-scatter of the resident segments, one 4-byte all-reduce for the global norm, fused AdamW on the slices (the bf16
    output lands in the parameter shard; sharded layers are NOT gathered here -- the next forward pass does that), all-gather of
    the resident segments only.

Per-GPU memory at LLaMA-3-70B (70.55 B parameters: 80 layers x 855.6 M + 2.1 B embeddings / lm_head + heads), world 8:
parameter shards 17.1 GB + gradient shards 17.1 GB + fp32 master / m / v 105.8 GB + resident 2 x 4.3 GB + 3 parameter slots and
2 gradient slots x 1.71 GB = 8.6 GB + frozen SO400M tower 0.9 GB = 158 GB, leaving 130 GB for activations: with per-layer recompute
(`gradient_checkpointing`, 134 MB per 4096-token sample and layer) more than enough for the reference's 4096-token micro-batches.

world_size 1 degenerates to copies (shard -> slot); it exists so that the whole mechanism runs on the single-GPU test box.
`shard_update` / `sumsq` / `clip_coef` / `accumulate` are injectable for the gloo CPU tests, like in Zero2AdamW.
"""
from __future__ import annotations


import torch
import torch.distributed as dist

from .zero2 import ALIGN, BF16, _MODE, _hip_clip_coef, _hip_shard_update, _hip_sumsq, tag_segments  # noqa: F401  (re-exported)


def _hip_accumulate(dst, src, first):
    """dst (+)= src on bf16 buffers (first micro-step of a window: plain copy)."""
    from . import ops
    ops.axpy_(dst, src, None, 1.0, not first)


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


class _Slot:
    def __init__(self, buf):
        self.buf = buf
        self.seg = None          # segment index currently held
        self.work = None         # pending collective writing to / reading from the buffer
        self.stamp = 0           # last use (LRU victim choice)


class Zero3AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, process_group=None,
                 shard_update=None, sumsq=None, clip_coef=None, accumulate=None, param_slots=3, grad_slots=2, min_shard_numel=1 << 20,
                 force_collectives=None, tensor_collectives=None):
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [dict(g, params=[p for p in g["params"] if p.requires_grad]) for g in params]
            groups = [g for g in groups if g["params"]]
        else:
            groups = [dict(params=[p for p in params if p.requires_grad])]
        if not groups or not groups[0]["params"]:
            raise ValueError("Zero3AdamW: no trainable parameters")
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(self.pg) if self.distributed else 1
        self.rank = dist.get_rank(self.pg) if self.distributed else 0
        force = _MODE["force_collectives"] if force_collectives is None else bool(force_collectives)
        tensor = _MODE["tensor_collectives"] if tensor_collectives is None else bool(tensor_collectives)
        self._coll = self.distributed and (self.world > 1 or force)       # (zero2.set_collective_mode: world-1 RCCL call pattern, gloo tests)
        self._tensor_coll = self._coll and (dist.get_backend(self.pg) == "nccl" or tensor)
        self.max_grad_norm = max_grad_norm
        self._shard_update = shard_update or _hip_shard_update
        self._sumsq = sumsq or _hip_sumsq
        self._clip_coef = clip_coef or _hip_clip_coef
        self._accumulate = accumulate or _hip_accumulate
        self._step = 0
        self._n_param_slots, self._n_grad_slots = max(2, int(param_slots)), max(1, int(grad_slots))
        # a decoder-layer run smaller than this stays resident (parameter groups cut a layer into runs: with the reference's decay /
        # no-decay groups the two norm weights of a layer form their own 2 x h run -- not worth a full-size slot and a collective)
        self._min_shard = int(min_shard_numel)
        self._layout([p for g in self.param_groups for p in g["params"]])
        self._hooked = False

    # ------------------------------------------------------------------ layout
    def _layout(self, params):
        dev, dt = params[0].device, params[0].dtype
        if any(p.dtype != dt or p.device != dev for p in params):
            raise ValueError("Zero3AdamW needs all trainable parameters on one device in one dtype")
        self.params = params
        chunk = ALIGN * self.world
        runs = []
        for p in params:
            key = (getattr(p, "_mm_segment", None), self._group_of[id(p)])
            if runs and runs[-1][0] == key:
                runs[-1][1].append(p)
            else:
                runs.append((key, [p]))
        self.total = sum(p.numel() for p in params)
        segs, so = [], 0
        for key, ps in runs:
            n_raw = sum(p.numel() for p in ps)
            n = (n_raw + chunk - 1) // chunk * chunk
            m = n // self.world
            offs, pos = [], 0
            for p in ps:
                offs.append(pos)
                pos += p.numel()
            segs.append({"key": key[0], "group": key[1], "params": ps, "offs": offs, "shapes": [tuple(p.shape) for p in ps], "n": n, "m": m,
                         "so": so, "sharded": key[0] is not None and n_raw >= self._min_shard})
            so += m
        self.segs = segs
        self.shard = so
        self.seg_of_key = {}                                # layer key -> its SHARDED segments
        for i, sg in enumerate(segs):
            if sg["sharded"]:
                self.seg_of_key.setdefault(sg["key"], []).append(i)
        self.layer_order = [i for i, sg in enumerate(segs) if sg["sharded"]]           # forward walk order
        max_n = max([segs[i]["n"] for i in self.layer_order], default=0)
        # fp32 master / moments of this rank's slices, in segment order
        self.master = torch.empty(self.shard, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        for sg in segs:
            full = torch.zeros(sg["n"], device=dev, dtype=dt)
            for p, o in zip(sg["params"], sg["offs"]):
                full[o:o + p.numel()].copy_(p.data.reshape(-1))
            lo = self.rank * sg["m"]
            self.master[sg["so"]:sg["so"] + sg["m"]].copy_(full[lo:lo + sg["m"]].float())
            if sg["sharded"]:
                sg["p_shard"] = full[lo:lo + sg["m"]].clone()
                sg["g_shard"] = torch.zeros(sg["m"], device=dev, dtype=dt)
                sg["g_live"] = False                       # does g_shard hold gradients of the current accumulation window?
                for p in sg["params"]:
                    p.data = torch.empty(0, device=dev, dtype=dt)      # released: (re)materialised by ensure_params()
                    p._mm_grad_buf = None
                    p.grad = None
                del full
            else:                                           # resident: the ZeRO-2 arrangement
                sg["param"] = full
                sg["grad"] = torch.zeros(sg["n"], device=dev, dtype=dt)
                sg["my_param"] = full[lo:lo + sg["m"]]
                sg["my_grad"] = sg["grad"][lo:lo + sg["m"]]
                for p, o in zip(sg["params"], sg["offs"]):
                    p.data = full[o:o + p.numel()].view(p.shape)
                    p._mm_grad_buf = sg["grad"][o:o + p.numel()].view(p.shape)
                    p.grad = None
        self._pslots = [_Slot(torch.empty(max_n, device=dev, dtype=dt)) for _ in range(self._n_param_slots if max_n else 0)]
        self._gslots = [_Slot(torch.empty(max_n, device=dev, dtype=dt)) for _ in range(self._n_grad_slots if max_n else 0)]
        self._pslot_of, self._gslot_of = {}, {}             # segment index -> slot
        self._p_rr = self._g_rr = 0
        self._norm_buf = torch.zeros(1, device=dev, dtype=torch.float32)
        self._coef = torch.ones(1, device=dev, dtype=torch.float32)

    # ------------------------------------------------------------------ parameter gather
    def _point_params(self, sg, buf):
        for p, o, shp in zip(sg["params"], sg["offs"], sg["shapes"]):
            n = 1
            for d in shp:
                n *= d
            p.data = buf[o:o + n].view(shp)

    def _issue_gather(self, i, protect=()):
        """Start materialising segment i in a parameter slot (no-op if it is already there or in flight)."""
        self._p_rr += 1
        if i in self._pslot_of:
            self._pslot_of[i].stamp = self._p_rr
            return self._pslot_of[i]
        # victim: a free slot, else the least recently used one that is not protected (the layer being computed right now)
        cands = [sl for sl in self._pslots if sl.seg is None] or [sl for sl in self._pslots if sl.seg not in protect]
        if not cands:                                        # a layer with more sharded runs than slots: grow the pool
            cands = [_Slot(torch.empty_like(self._pslots[0].buf))]
            self._pslots.append(cands[0])
        slot = min(cands, key=lambda sl: sl.stamp)
        slot.stamp = self._p_rr
        if slot.seg is not None:
            self._pslot_of.pop(slot.seg, None)
            old = self.segs[slot.seg]
            for p in old["params"]:                          # whoever reads a released layer without params_ready() fails loudly
                p.data = torch.empty(0, device=slot.buf.device, dtype=slot.buf.dtype)
        if slot.work is not None:
            slot.work.wait()
            slot.work = None
        sg = self.segs[i]
        full = slot.buf[:sg["n"]]
        if self._coll:
            if self._tensor_coll:
                mine = full[self.rank * sg["m"]:(self.rank + 1) * sg["m"]]
                mine.copy_(sg["p_shard"])
                slot.work = dist.all_gather_into_tensor(full, mine, group=self.pg, async_op=True)
            else:                                            # gloo (CPU tests)
                parts = [torch.empty_like(sg["p_shard"]) for _ in range(self.world)]
                dist.all_gather(parts, sg["p_shard"], group=self.pg)
                for r, t in enumerate(parts):
                    full[r * sg["m"]:(r + 1) * sg["m"]].copy_(t)
        else:
            full[:sg["m"]].copy_(sg["p_shard"])
        slot.seg = i
        self._pslot_of[i] = slot
        self._point_params(sg, slot.buf)
        return slot

    def ensure_params(self, key, backward=False):
        """The parameters of decoder layer `key` are complete on the current stream when this returns; the neighbour the walk
        reaches next (forward: the following layer, backward: the preceding one) starts gathering."""
        idxs = self.seg_of_key.get(key, ())
        for i in idxs:
            slot = self._issue_gather(i, protect=idxs)
            if slot.work is not None:
                slot.work.wait()                             # current stream waits for the all-gather
                slot.work = None
        if idxs and len(self._pslots) > len(idxs):
            pos = self.layer_order.index(idxs[-1] if not backward else idxs[0])
            nxt = pos + (-1 if backward else 1)
            if 0 <= nxt < len(self.layer_order):
                self._issue_gather(self.layer_order[nxt], protect=idxs)
        if backward:
            for i in idxs:
                self._attach_grad_slot(i)

    # ------------------------------------------------------------------ gradient slots
    def _attach_grad_slot(self, i):
        if i in self._gslot_of:
            return
        free = [sl for sl in self._gslots if sl.seg is None]
        if not free:                                         # more runs of one layer in flight than slots: grow the pool
            free = [_Slot(torch.empty_like(self._gslots[0].buf))]
            self._gslots.append(free[0])
        slot = min(free, key=lambda sl: sl.stamp)
        self._g_rr += 1
        slot.stamp = self._g_rr
        if slot.work is not None:                            # the reduce-scatter that last read this buffer
            slot.work()
            slot.work = None
        sg = self.segs[i]
        slot.seg = i
        self._gslot_of[i] = slot
        n_raw = sum(_numel(shp) for shp in sg["shapes"])
        if n_raw < sg["n"]:
            slot.buf[n_raw:sg["n"]].zero_()                  # alignment padding: reduced and norm-ed with the rest, must be zero
        for p, o, shp in zip(sg["params"], sg["offs"], sg["shapes"]):
            n = 1
            for d in shp:
                n *= d
            p._mm_grad_buf = slot.buf[o:o + n].view(shp)
            p.grad = None

    def layer_backward_done(self, key):
        """All gradients of decoder layer `key` are in its gradient slot: reduce-scatter, fold this rank's slice into the shard."""
        for i in self.seg_of_key.get(key, ()):
            sg = self.segs[i]
            slot = self._gslot_of.pop(i, None)
            if slot is None:
                continue
            for p in sg["params"]:
                if p.grad is None and p._mm_grad_buf is not None:
                    p._mm_grad_buf.zero_()                   # a parameter without gradient this micro-step contributes zeros
                elif p.grad is not None and p.grad.data_ptr() != p._mm_grad_buf.data_ptr():
                    p._mm_grad_buf.copy_(p.grad)
            full = slot.buf[:sg["n"]]
            mine = full[self.rank * sg["m"]:(self.rank + 1) * sg["m"]]
            first = not sg["g_live"]
            sg["g_live"] = True
            if self._tensor_coll:
                w = dist.reduce_scatter_tensor(mine, full, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

                def finish(w=w, sg=sg, mine=mine, first=first):
                    w.wait()
                    self._accumulate(sg["g_shard"], mine, first)
                slot.work = finish
            elif self._coll:                                 # gloo: all_reduce in fp32 stands in for reduce-scatter
                g32 = full.float()
                dist.all_reduce(g32, op=dist.ReduceOp.SUM, group=self.pg)
                self._accumulate(sg["g_shard"], g32[self.rank * sg["m"]:(self.rank + 1) * sg["m"]].to(full.dtype), first)
            else:
                self._accumulate(sg["g_shard"], mine, first)
            slot.seg = None
            for p in sg["params"]:
                p.grad = None
                p._mm_grad_buf = None

    def _drain_grad_slots(self):
        for slot in self._gslots:
            if slot.work is not None:
                slot.work()
                slot.work = None

    # ------------------------------------------------------------------ hooks
    def enable_hooks(self):
        """Route the model's announcements to this optimizer: "about to read layer X" (forward / decode / backward) -> gather,
        "layer X's backward is finished" -> reduce-scatter."""
        from . import functional as F

        mine = {id(p) for p in self.params}

        def owns(layer):                                     # the hooks are process-wide: ignore layers of any other model
            p0 = next(layer.parameters(), None) if layer is not None else None
            return p0 is not None and id(p0) in mine

        def ready(layer, backward=False):
            if owns(layer):
                self.ensure_params(getattr(layer, "_mm_segment", None), backward=backward)

        def done(layer):
            if owns(layer):
                self.layer_backward_done(getattr(layer, "_mm_segment", None))
        F.set_param_ready_hook(ready)
        F.set_layer_grad_hook(done)
        self._hooked = True
        return self

    enable_overlap = enable_hooks                            # same call site as Zero2AdamW (bench / trainer)

    def arm_overlap(self):
        return self

    # ------------------------------------------------------------------ step
    def zero_grad(self, set_to_none: bool = True):
        for sg in self.segs:
            for p in sg["params"]:
                p.grad = None
            if sg["sharded"]:
                sg["g_live"] = False

    def _hyper(self, gi):
        g = self.param_groups[gi]
        b1, b2 = g["betas"]
        return (float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], self._step, self._coef)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        self._drain_grad_slots()
        for i in list(self._gslot_of):                       # a layer whose backward never announced itself (hooks not installed)
            self.layer_backward_done(self.segs[i]["key"])
        self._drain_grad_slots()
        # resident segments: ZeRO-2 reduce-scatter
        for sg in self.segs:
            if sg["sharded"]:
                continue
            for p in sg["params"]:
                if p.grad is None:
                    p._mm_grad_buf.zero_()
                elif p.grad.data_ptr() != p._mm_grad_buf.data_ptr():
                    p._mm_grad_buf.copy_(p.grad)
            if self._coll:
                if self._tensor_coll:
                    dist.reduce_scatter_tensor(sg["my_grad"], sg["grad"], op=dist.ReduceOp.SUM, group=self.pg)
                else:
                    g32 = sg["grad"].float()
                    dist.all_reduce(g32, op=dist.ReduceOp.SUM, group=self.pg)
                    sg["grad"].copy_(g32)
        self._norm_buf.zero_()
        for sg in self.segs:
            if sg["sharded"] and not sg["g_live"]:
                sg["g_shard"].zero_()
            self._sumsq(sg["g_shard"] if sg["sharded"] else sg["my_grad"], self._norm_buf)
        if self._coll:
            dist.all_reduce(self._norm_buf, op=dist.ReduceOp.SUM, group=self.pg)
        inv_world = 1.0 / self.world
        mx = self.max_grad_norm / inv_world if self.max_grad_norm and self.max_grad_norm > 0 else 0.0
        self._clip_coef(self._norm_buf, mx, inv_world, self._coef)
        self._step += 1
        for sg in self.segs:
            so, m = sg["so"], sg["m"]
            g, p_out = (sg["g_shard"], sg["p_shard"]) if sg["sharded"] else (sg["my_grad"], sg["my_param"])
            self._shard_update(self.master[so:so + m], self.exp_avg[so:so + m], self.exp_avg_sq[so:so + m], g, p_out, *self._hyper(sg["group"]))
        # resident parameters are all-gathered now; sharded layers at their next use (their slots hold stale copies: drop them)
        self._all_gather_resident()
        self.release_params()
        for sg in self.segs:
            if sg["sharded"]:
                sg["g_live"] = False
        if self.master.is_cuda:
            from . import functional as F
            F.bump_param_generation()
        return None

    def _all_gather_resident(self):
        for sg in self.segs:
            if sg["sharded"] or not self._coll:
                continue
            if self._tensor_coll:
                dist.all_gather_into_tensor(sg["param"], sg["my_param"], group=self.pg)
            else:
                parts = [torch.empty_like(sg["my_param"]) for _ in range(self.world)]
                dist.all_gather(parts, sg["my_param"].clone(), group=self.pg)
                for r, t in enumerate(parts):
                    sg["param"][r * sg["m"]:(r + 1) * sg["m"]].copy_(t)

    def release_params(self):
        """Forget every gathered layer (after an update, or to free nothing but bookkeeping: the slots themselves persist)."""
        for slot in self._pslots:
            if slot.work is not None:
                slot.work.wait()
                slot.work = None
            if slot.seg is not None:
                for p in self.segs[slot.seg]["params"]:
                    p.data = torch.empty(0, device=slot.buf.device, dtype=slot.buf.dtype)
                slot.seg = None
        self._pslot_of = {}

    def synchronize(self):
        """Everything this optimizer started has finished: gradient reductions folded in, prefetched all-gathers complete (a gather that
        nobody consumed -- the neighbour prefetched by the last ensure_params() -- must not be in flight when the process group goes away)."""
        self._drain_grad_slots()
        for slot in self._pslots:
            if slot.work is not None:
                slot.work.wait()
                slot.work = None
        if self.master.is_cuda:
            torch.cuda.current_stream().synchronize()

    def grad_norm_value(self):
        import math
        return math.sqrt(float(self._norm_buf)) / self.world

    # ------------------------------------------------------------------ full parameters (model save, evaluation without hooks)
    @torch.no_grad()
    def iter_full_layers(self):
        """Yields (segment, full) for every sharded layer segment in order, `full` a flat bf16 view holding the complete layer.  ONE
        layer-sized buffer is reused for all layers (peak = a single 1.7 GB slot at LLaMA-3-70B, not 80 of them): the consumer must copy
        what it keeps before advancing.  Collective -- every rank of the group must iterate."""
        buf = None
        for i in self.layer_order:
            sg = self.segs[i]
            if buf is None or buf.numel() < sg["n"]:
                buf = torch.empty(max(self.segs[j]["n"] for j in self.layer_order), device=sg["p_shard"].device, dtype=sg["p_shard"].dtype)
            full = buf[:sg["n"]]
            if self._coll:
                if self._tensor_coll:
                    dist.all_gather_into_tensor(full, sg["p_shard"], group=self.pg)
                else:
                    parts = [torch.empty_like(sg["p_shard"]) for _ in range(self.world)]
                    dist.all_gather(parts, sg["p_shard"], group=self.pg)
                    for r, t in enumerate(parts):
                        full[r * sg["m"]:(r + 1) * sg["m"]].copy_(t)
                    del parts
            else:
                full[:sg["m"]].copy_(sg["p_shard"])
            yield sg, full

    @torch.no_grad()
    def gather_full_parameters(self, device=None, keep=True):
        """{parameter: full bf16 tensor} of the sharded layers (`stage3_gather_16bit_weights_on_model_save`, reference
        scripts/zero3.json:26).  Gathered ONE LAYER AT A TIME through a single reusable buffer; every tensor is copied out to `device`
        (default: the shard's device; "cpu" for a model save) before the next layer is gathered.  keep=False: take part in the
        collectives but keep nothing (the ranks that do not write the checkpoint)."""
        out = {}
        for sg, full in self.iter_full_layers():
            if not keep:
                continue
            for p, o, shp in zip(sg["params"], sg["offs"], sg["shapes"]):
                n = 1
                for d in shp:
                    n *= d
                t = full[o:o + n].view(shp)
                u = t if device is None else t.to(device)
                out[p] = u.clone() if u.data_ptr() == t.data_ptr() else u     # never alias the reusable gather buffer ("cuda" vs "cuda:0" compare unequal)
        return out

    # ------------------------------------------------------------------ checkpointing of the rank's shard
    def state_dict(self):
        self.synchronize()
        return {"step": self._step, "master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "world": self.world,
                "rank": self.rank, "total": self.total, "zero_stage": 3,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if sd["world"] != self.world or sd["total"] != self.total:
            raise ValueError("Zero3AdamW shard checkpoint was written with a different world size / parameter set")
        self._step = sd["step"]
        self.master.copy_(sd["master"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        # the bf16 parameters follow the restored master weights
        for sg in self.segs:
            dst = sg["p_shard"] if sg["sharded"] else sg["my_param"]
            dst.copy_(self.master[sg["so"]:sg["so"] + sg["m"]].to(dst.dtype))
        self._all_gather_resident()                         # the other ranks' slices of the resident tensors
        self.release_params()
        if self.master.is_cuda:
            from . import functional as F
            F.bump_param_generation()

    def full_state_dict(self, model, device=None, keep=True):
        """model.state_dict() with the sharded decoder layers gathered (`stage3_gather_16bit_weights_on_model_save`): what
        `save_pretrained` must be given under ZeRO-3, where the module tree itself holds no layer weights between steps.
        COLLECTIVE: every rank must call it (one all-gather per sharded layer); ranks with keep=False get None.  device="cpu" moves
        each layer to the host as it arrives, so the device peak is one layer, not the model."""
        full = self.gather_full_parameters(device=device, keep=keep)
        if not keep:
            return None
        by_id = {id(p): t for p, t in full.items()}
        out = {}
        names = {id(p): n for n, p in model.named_parameters()}
        for name, t in model.state_dict(keep_vars=True).items():
            t = by_id.get(id(t), t).detach()
            out[name] = t.to(device) if device is not None else t
        assert all(names[i] in out for i in by_id)
        return out

    def memory_report(self):
        """Bytes held by this rank, by role."""
        e = self.master.element_size()
        b = 2
        shards = sum(sg["m"] for sg in self.segs if sg["sharded"])
        resident = sum(sg["n"] for sg in self.segs if not sg["sharded"])
        slot_n = self._pslots[0].buf.numel() if self._pslots else 0
        return {"param_shards": shards * b, "grad_shards": shards * b, "resident_params": resident * b, "resident_grads": resident * b,
                "optimizer_fp32": 3 * self.shard * e, "param_slots": len(self._pslots) * slot_n * b, "grad_slots": len(self._gslots) * slot_n * b}
