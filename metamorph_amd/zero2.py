"""ZeRO-2 style data-parallel optimizer over RCCL (one process per GPU).

Replaces the reference's DeepSpeed path (reference scripts/zero2.json:16-24 driven by HF Trainer with
optim="adamw_torch", train.py:82):

  * all trainable parameters live in ONE flat bf16 buffer, their gradients in a second flat bf16 buffer
    (`contiguous_gradients`); the backward kernels write weight gradients straight into it;
  * the flat space is cut into SEGMENTS (one per decoder layer + the runs of parameters in between: a few hundred MB
    each -- xGMI is point-to-point, so few large collectives beat many small ones); a segment's gradients are
    reduce-scattered in place (sum; the 1/world mean is folded into the update kernel) as soon as the backward of its
    decoder layer has finished, asynchronously on RCCL's stream, while the backward of the earlier layers keeps the
    compute stream busy (`overlap_comm` of scripts/zero2.json:20); whatever is left goes out at step();
  * every rank owns slice `rank` of every segment: fp32 master weights + Adam moments for those slices only
    (12 B/param sharded), updated by the fused HIP kernel (mm355_adamw_shard) that also emits the bf16 weights;
  * global grad-norm clipping costs one 4-byte all-reduce;
  * updated bf16 shards are all-gathered back into the flat parameter buffer.

The data path has no other collective: samples are independent (CE / cosine losses are means over the local
micro-batch, reference metamorph_llama.py:407-413,453), so per-rank work is fixed as ranks are added (weak scaling).

`shard_update` / `sumsq` are injectable so that the partition + collective logic can be exercised on CPU with the
gloo backend in tests (the product default is the HIP kernels; there is no CPU fallback in the product).
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist

BF16 = torch.bfloat16
ALIGN = 256            # elements; keeps every shard boundary 512-byte aligned


def _hip_shard_update(p32, m, v, g, p_out, lr, b1, b2, eps, wd, step, scale_dev):
    from . import ops
    ops.adamw_shard_(p32, m, v, g, p_out, lr, b1, b2, eps, wd, step, scale_dev)


def _hip_sumsq(x, out):
    from . import ops
    ops.sumsq_(x, out)


def _hip_clip_coef(sumsq, max_norm, pre, out):
    from . import ops
    ops.clip_coef(sumsq, max_norm, pre, out)



def tag_segments(model):
    """Mark the parameters of every decoder layer as one reduction segment (call BEFORE constructing Zero2AdamW).  The
    layer module carries the same key so that DecoderLayerFn.backward can announce "this layer's gradients are final"."""
    inner = model.get_model() if hasattr(model, "get_model") else model
    for i, layer in enumerate(getattr(inner, "layers", [])):
        layer._mm_segment = ("layer", i)
        for p in layer.parameters():
            p._mm_segment = ("layer", i)


# How the exchange is driven when nothing else says so (constructor arguments override; tests / bench flip the module defaults through
# set_collective_mode() -- there is no environment switch):
#   force_collectives   run the collectives at world size 1 too (exercises the RCCL call pattern -- in-place reduce-scatter / all-gather,
#                       async handles, side streams -- on a single-GPU box)
#   tensor_collectives  use the RCCL form (in-place reduce_scatter_tensor / all_gather_into_tensor on slices of the flat buffers) on a
#                       backend other than nccl (the gloo tests drive this branch at world 2-8 on CPU)
_MODE = {"force_collectives": False, "tensor_collectives": False}


def set_collective_mode(force_collectives=None, tensor_collectives=None):
    """Change the module defaults; returns the previous ones (pass them back to restore)."""
    old = dict(_MODE)
    if force_collectives is not None:
        _MODE["force_collectives"] = bool(force_collectives)
    if tensor_collectives is not None:
        _MODE["tensor_collectives"] = bool(tensor_collectives)
    return old


class _Pending:
    """One in-flight segment reduction (RCCL reduce-scatter, or the all-reduce stand-in of the gloo tests)."""

    def __init__(self, work, finish=None):
        self.work, self.finish = work, finish

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if self.finish is not None:
            self.finish()


class Zero2AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 process_group=None, shard_update=None, sumsq=None, clip_coef=None, overlap=True, async_update=False,
                 force_collectives=None, tensor_collectives=None):
        # `params`: an iterable of tensors, or of torch-style group dicts ({"params": [...], "lr": ..., "weight_decay": ...}) --
        # e.g. the reference's separate `vision_lr` group for the tower (metamorph_trainer.py:201-233)
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [dict(g, params=[p for p in g["params"] if p.requires_grad]) for g in params]
            groups = [g for g in groups if g["params"]]
        else:
            groups = [dict(params=[p for p in params if p.requires_grad])]
        if not groups or not groups[0]["params"]:
            raise ValueError("Zero2AdamW: no trainable parameters")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(groups, defaults)
        self._group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        params = [p for g in self.param_groups for p in g["params"]]
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(self.pg) if self.distributed else 1
        self.rank = dist.get_rank(self.pg) if self.distributed else 0
        self.max_grad_norm = max_grad_norm
        force = _MODE["force_collectives"] if force_collectives is None else bool(force_collectives)
        tensor = _MODE["tensor_collectives"] if tensor_collectives is None else bool(tensor_collectives)
        # collectives are skipped at world size 1 unless forced (see set_collective_mode)
        self._coll = self.distributed and (self.world > 1 or force)
        # the RCCL form of the exchange: in-place reduce_scatter_tensor / all_gather_into_tensor on slices of the flat buffers
        self._tensor_coll = self._coll and (dist.get_backend(self.pg) == "nccl" or tensor)
        self._shard_update = shard_update or _hip_shard_update
        self._sumsq = sumsq or _hip_sumsq
        self._clip_coef = clip_coef or _hip_clip_coef
        self.overlap = True if overlap is None else bool(overlap)
        self._step = 0
        self._armed = False
        self._pending = {}                 # segment index -> _Pending
        self._flatten(params)
        # asynchronous update (opt-in, async_update=True): the per-segment AdamW kernels and all-gathers run on a side stream
        # behind step(); the next forward pass waits per segment right before it reads that segment's parameters
        # (functional.params_ready).  On ONE GPU it buys nothing (measured: the HBM-bound update slows the concurrent GEMMs by
        # as much as it hides, 1014.9 vs 1012.7 ms/step); its purpose is hiding the all-gather at world > 1, which this round
        # could not measure, hence off by default.
        on_gpu = self.flat_param.is_cuda
        self.async_update = on_gpu and bool(async_update)
        self._upd_stream = torch.cuda.Stream(device=self.flat_param.device) if self.async_update else None
        self._ready = {}                   # segment index -> event recorded on the update stream
        self._waited = set()
        self._async_hooked = False
        # comm_timing = True: HIP events on the compute stream around the parts of step() that WAIT for collectives (bench.py at
        # world > 1): what the step pays for communication after overlap, per phase
        self.comm_timing = False
        self._comm_events = {"reduce_scatter_exposed": [], "norm_all_reduce": [], "all_gather": []}

    def set_async_update(self, on):
        """Switch the asynchronous update / all-gather (see above) on an existing optimizer (bench.py's A/B at world > 1)."""
        self.wait_all()
        on = bool(on) and self.flat_param.is_cuda
        if on and self._upd_stream is None:
            self._upd_stream = torch.cuda.Stream(device=self.flat_param.device)
        self.async_update = on
        if on:
            self.enable_async_wait()
        elif self._async_hooked:
            from . import functional as F
            F.set_param_ready_hook(None)
            self._async_hooked = False
        return self

    def _timed(self, phase):
        """Context manager: HIP events around a phase on the current stream when comm_timing is on."""
        opt = self

        class _T:
            def __enter__(self_):
                if opt.comm_timing and opt.flat_param.is_cuda:
                    self_.s = torch.cuda.Event(enable_timing=True)
                    self_.s.record()
                return self_

            def __exit__(self_, *a):
                if opt.comm_timing and opt.flat_param.is_cuda:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    opt._comm_events[phase].append((self_.s, e))
        return _T()

    def comm_summary(self, steps):
        """{phase: ms per step} from the recorded events (syncs); clears them."""
        if self.flat_param.is_cuda:
            torch.cuda.synchronize()
        out = {k: round(sum(s.elapsed_time(e) for s, e in v) / max(steps, 1), 3) for k, v in self._comm_events.items()}
        self._comm_events = {k: [] for k in self._comm_events}
        return out

    def comm_bytes_per_step(self):
        """Payload of one optimizer step's exchange, per rank: in-place reduce-scatter of the bf16 gradient buffer, all-gather of the bf16
        parameter buffer (both `padded` elements, cut into `segments` collectives), one fp32 norm all-reduce."""
        e = self.flat_param.element_size()
        return {"reduce_scatter_bytes": int(self.padded) * e, "all_gather_bytes": int(self.padded) * e, "segments": len(self.segs),
                "norm_all_reduce_bytes": 4, "world": self.world}

    # ------------------------------------------------------------------ layout
    def _flatten(self, params):
        dev, dt = params[0].device, params[0].dtype
        if any(p.dtype != dt or p.device != dev for p in params):
            raise ValueError("Zero2AdamW needs all trainable parameters on one device in one dtype")
        self.params = params
        chunk = ALIGN * self.world
        # consecutive parameters with the same `_mm_segment` key form a segment; NO padding inside a segment (fused q/k/v and
        # gate/up stay contiguous), every segment is padded to a multiple of ALIGN * world so that all slices stay aligned
        runs = []                                            # a segment never spans two parameter groups (their lr may differ)
        for p in params:
            key = (getattr(p, "_mm_segment", None), self._group_of[id(p)])
            if runs and runs[-1][0] == key:
                runs[-1][1].append(p)
            else:
                runs.append((key, [p]))
        self.total = sum(p.numel() for p in params)
        offs, segs, pos = {}, [], 0
        for key, ps in runs:
            lo = pos
            for p in ps:
                offs[id(p)] = pos
                pos += p.numel()
            n = (pos - lo + chunk - 1) // chunk * chunk
            pos = lo + n
            segs.append({"key": key[0], "group": key[1], "lo": lo, "n": n, "params": ps})
        self.padded = pos
        self.shard = self.padded // self.world
        self.flat_param = torch.zeros(self.padded, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(self.padded, device=dev, dtype=dt)
        for p in params:
            o = offs[id(p)]
            view = self.flat_param[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p._mm_grad_buf = self.flat_grad[o:o + p.numel()].view(p.shape)
            p.grad = None
        self.offsets = [offs[id(p)] for p in params]
        so = 0
        for sg in segs:                    # this rank's slice of every segment, and where it sits in the shard arrays
            m = sg["n"] // self.world
            lo = sg["lo"] + self.rank * m
            sg.update(m=m, so=so, grad=self.flat_grad[sg["lo"]:sg["lo"] + sg["n"]], param=self.flat_param[sg["lo"]:sg["lo"] + sg["n"]],
                      my_grad=self.flat_grad[lo:lo + m], my_param=self.flat_param[lo:lo + m])
            so += m
        self.segs = segs
        self.seg_of_key = {}               # key -> segment indices (more than one when parameter groups cut through a key)
        for i, sg in enumerate(segs):
            if sg["key"] is not None:
                self.seg_of_key.setdefault(sg["key"], []).append(i)
        self.master = torch.cat([sg["my_param"] for sg in segs]).float()
        self.exp_avg = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self._norm_buf = torch.zeros(1, device=dev, dtype=torch.float32)
        self._coef = torch.ones(1, device=dev, dtype=torch.float32)
        self.grad_norm = None

    # ------------------------------------------------------------------ collectives
    @staticmethod
    def _settle_grads(ps):
        """Parameters that received no gradient this step contribute zeros; a gradient produced outside the flat buffer
        (e.g. by autograd) is copied in."""
        for p in ps:
            if p.grad is None:
                p._mm_grad_buf.zero_()
            elif p.grad.data_ptr() != p._mm_grad_buf.data_ptr():
                p._mm_grad_buf.copy_(p.grad)

    def _launch_reduce(self, i, async_op):
        """Sum segment i over the ranks; this rank's slice ends up in place (the other slices are don't-care afterwards)."""
        sg = self.segs[i]
        self._settle_grads(sg["params"])
        if self._tensor_coll:                             # RCCL: in-place reduce-scatter (output = input + rank * count)
            w = dist.reduce_scatter_tensor(sg["my_grad"], sg["grad"], op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
            self._pending[i] = _Pending(w if async_op else None)
        else:                                             # gloo (CPU tests): no reduce_scatter -> all_reduce in fp32
            g = sg["grad"].float() if sg["grad"].dtype == BF16 else sg["grad"]
            w = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
            fin = (lambda g=g, sg=sg: sg["grad"].copy_(g)) if g is not sg["grad"] else None
            self._pending[i] = _Pending(w if async_op else None, fin)

    def arm_overlap(self):
        """Call before the backward pass whose gradients are final (the last micro-step of an accumulation window): from
        now until step(), notify_segment_ready() starts that segment's reduction right away."""
        self.wait_all()                                      # the previous update has consumed the gradients before they are rewritten
        self._armed = self.overlap and self._coll
        return self

    def notify_segment_ready(self, key):
        if not self._armed:
            return
        for i in self.seg_of_key.get(key, ()):
            if i not in self._pending:
                self._launch_reduce(i, async_op=True)

    def enable_overlap(self):
        """Route DecoderLayerFn's "layer gradients are final" announcements to this optimizer."""
        from . import functional as F
        mine = {id(p) for p in self.params}

        def done(layer):                                     # the hook is process-wide: ignore layers of any other model
            p0 = next(layer.parameters(), None)
            if p0 is not None and id(p0) in mine:
                self.notify_segment_ready(getattr(layer, "_mm_segment", None))
        F.set_layer_grad_hook(done)
        if self.async_update:
            self.enable_async_wait()
        return self

    def _reduce_grads(self):
        self.wait_all()
        if not self._coll:
            self._settle_grads(self.params)
            return
        with self._timed("reduce_scatter_exposed"):          # what is left of the reductions once backward has ended
            for i in range(len(self.segs)):
                if i not in self._pending:
                    self._launch_reduce(i, async_op=False)
            for pend in self._pending.values():
                pend.wait()
        self._pending = {}
        self._armed = False

    def _all_gather_params(self):
        if not self._coll:
            return
        if self._tensor_coll:
            works = [dist.all_gather_into_tensor(sg["param"], sg["my_param"], group=self.pg, async_op=True) for sg in self.segs]
            for w in works:
                w.wait()
        else:
            for sg in self.segs:
                parts = [torch.empty_like(sg["my_param"]) for _ in range(self.world)]
                dist.all_gather(parts, sg["my_param"].clone(), group=self.pg)
                for r, t in enumerate(parts):
                    sg["param"][r * sg["m"]:(r + 1) * sg["m"]].copy_(t)

    def _my_slices(self):
        """(shard offset, length, gradient slice, parameter slice, group) runs of this rank; one run when the slices are
        adjacent and share their hyper-parameters."""
        if self.world == 1 and len(self.param_groups) == 1:
            return [(0, self.padded, self.flat_grad, self.flat_param, 0)]
        return [(sg["so"], sg["m"], sg["my_grad"], sg["my_param"], sg["group"]) for sg in self.segs]

    def _hyper(self, gi):
        g = self.param_groups[gi]
        b1, b2 = g["betas"]
        return (float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], self._step, self._coef)

    # ------------------------------------------------------------------ asynchronous update
    def _update_order(self):
        """Segments outside the decoder layers first (embeddings, final norm, projector, heads: the next forward pass reads
        them first and last), then the decoder layers in forward order."""
        idx = list(range(len(self.segs)))
        return [i for i in idx if self.segs[i]["key"] is None] + [i for i in idx if self.segs[i]["key"] is not None]

    def _update_async(self):
        main = torch.cuda.current_stream()
        side = self._upd_stream
        side.wait_stream(main)                               # gradients, norm and clip coefficient are final
        self._ready, self._waited = {}, set()
        nccl = self._tensor_coll
        with torch.cuda.stream(side):
            prev = None                                      # (segment, all-gather work) one step behind the update kernels
            for i in self._update_order():
                sg = self.segs[i]
                so, m, gs, ps = sg["so"], sg["m"], sg["my_grad"], sg["my_param"]   # (world 1: the slice is the whole segment)
                self._shard_update(self.master[so:so + m], self.exp_avg[so:so + m], self.exp_avg_sq[so:so + m], gs, ps, *self._hyper(sg["group"]))
                work = None
                if self._coll:
                    if not nccl:
                        raise RuntimeError("asynchronous update needs the RCCL backend")
                    work = dist.all_gather_into_tensor(sg["param"], sg["my_param"], group=self.pg, async_op=True)
                if prev is not None:
                    self._finish_segment(*prev)
                prev = (i, work)
            if prev is not None:
                self._finish_segment(*prev)

    def _finish_segment(self, i, work):
        if work is not None:
            work.wait()                                      # the update stream waits for the all-gather of segment i
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._ready[i] = ev

    def wait_segment(self, key):
        """Make the current stream wait until the parameters of segment `key` (a `_mm_segment` key; None = every segment
        outside the decoder layers) carry the last step()'s update.  No-op when nothing is pending."""
        if not self._ready:
            return
        if key is None:
            todo = [i for i, sg in enumerate(self.segs) if sg["key"] is None]
        else:
            todo = self.seg_of_key.get(key, ())
        cur = torch.cuda.current_stream()
        for i in todo:
            if i not in self._waited and i in self._ready:
                cur.wait_event(self._ready[i])
                self._waited.add(i)

    def wait_all(self):
        """Current stream waits for every pending segment update (call before reading parameters or optimizer state outside a
        forward pass, and it is called before gradients are written again)."""
        if self._ready:
            cur = torch.cuda.current_stream()
            for i, ev in self._ready.items():
                if i not in self._waited:
                    cur.wait_event(ev)
            self._ready, self._waited = {}, set()

    def enable_async_wait(self):
        """Route the model's "about to read these parameters" announcements (functional.params_ready) to wait_segment."""
        from . import functional as F
        F.set_param_ready_hook(lambda layer, backward=False: self.wait_segment(None if layer is None else getattr(layer, "_mm_segment", None)))
        self._async_hooked = True
        return self

    # ------------------------------------------------------------------ step
    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    def synchronize(self):
        """Host-side barrier on everything step() started."""
        self.wait_all()
        if self.flat_param.is_cuda:
            torch.cuda.current_stream().synchronize()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        self._reduce_grads()
        inv_world = 1.0 / self.world
        # global L2 norm of the MEAN gradient: sqrt(sum over shards) * 1/world
        self._norm_buf.zero_()
        for _, _, g, _, _ in self._my_slices():
            self._sumsq(g, self._norm_buf)
        if self._coll:
            with self._timed("norm_all_reduce"):
                dist.all_reduce(self._norm_buf, op=dist.ReduceOp.SUM, group=self.pg)
        # coef = min(1, max_norm / (||mean grad|| + 1e-6)) * (1/world), computed on the device (no host sync)
        self._clip_coef_scaled(inv_world)
        self._step += 1
        if self.async_update:
            self._update_async()
            if not self._async_hooked:                       # nobody announces parameter reads: behave synchronously
                self.wait_all()
        else:
            for so, m, gs, ps, gi in self._my_slices():
                self._shard_update(self.master[so:so + m], self.exp_avg[so:so + m], self.exp_avg_sq[so:so + m], gs, ps, *self._hyper(gi))
            with self._timed("all_gather"):
                self._all_gather_params()
        self.grad_norm = self._norm_buf        # sum of squares of the summed gradient (device scalar); see grad_norm_value()
        if self.flat_param.is_cuda:
            from . import functional as F
            F.bump_param_generation()          # parameters changed behind torch's version counters: drop derived-operand caches
        return None

    def _clip_coef_scaled(self, inv_world):
        # ||mean|| = inv_world * sqrt(sumsq_of_sum).  clip_coef kernel computes min(1, max_norm/(sqrt(s)+1e-6))*pre
        # for s = sumsq of the SUM; rescale max_norm accordingly.
        mx = self.max_grad_norm / inv_world if self.max_grad_norm and self.max_grad_norm > 0 else 0.0
        self._clip_coef(self._norm_buf, mx, inv_world, self._coef)

    def grad_norm_value(self):
        """Host float of the last global gradient norm (syncs)."""
        return math.sqrt(float(self._norm_buf)) / self.world

    # ------------------------------------------------------------------ checkpointing of the rank's shard
    def state_dict(self):
        self.wait_all()
        return {"step": self._step, "master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "world": self.world, "rank": self.rank, "total": self.total, "param_groups": [
                    {k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if sd["world"] != self.world or sd["total"] != self.total:
            raise ValueError("Zero2AdamW shard checkpoint was written with a different world size / parameter set")
        self._step = sd["step"]
        self.master.copy_(sd["master"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        # the bf16 parameters follow the restored master weights, on every rank (like load_consolidated_optimizer_state)
        for sg in self.segs:
            sg["my_param"].copy_(self.master[sg["so"]:sg["so"] + sg["m"]].to(sg["my_param"].dtype))
        self._all_gather_params()
        if self.flat_param.is_cuda:
            from . import functional as F
            F.bump_param_generation()
