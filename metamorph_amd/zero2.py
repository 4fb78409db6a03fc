"""ZeRO-2 style data-parallel optimizer over RCCL (one process per GPU).

Replaces the reference's DeepSpeed path (reference scripts/zero2.json:16-24 driven by HF Trainer with
optim="adamw_torch", train.py:82):

  * all trainable parameters live in ONE flat bf16 buffer, their gradients in a second flat bf16 buffer
    (`contiguous_gradients`); the backward kernels write weight gradients straight into it;
  * after backward the flat gradient buffer is reduce-scattered (sum; the 1/world mean is folded into the
    update kernel) in large buckets -- xGMI is point-to-point, so few large collectives beat many small ones;
  * every rank owns 1/world of the flat space: fp32 master weights + Adam moments for that shard only
    (12 B/param sharded), updated by one fused HIP kernel (mm355_adamw_shard) that also emits the bf16 weights;
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



class Zero2AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 process_group=None, bucket_elems=0, shard_update=None, sumsq=None, clip_coef=None):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("Zero2AdamW: no trainable parameters")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(self.pg) if self.distributed else 1
        self.rank = dist.get_rank(self.pg) if self.distributed else 0
        self.max_grad_norm = max_grad_norm
        self.bucket_elems = int(bucket_elems)
        self._shard_update = shard_update or _hip_shard_update
        self._sumsq = sumsq or _hip_sumsq
        self._clip_coef = clip_coef or _hip_clip_coef
        self._step = 0
        self._flatten(params)

    # ------------------------------------------------------------------ layout
    def _flatten(self, params):
        dev, dt = params[0].device, params[0].dtype
        if any(p.dtype != dt or p.device != dev for p in params):
            raise ValueError("Zero2AdamW needs all trainable parameters on one device in one dtype")
        self.params = params
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += p.numel()            # NO padding between parameters: fused q/k/v and gate/up stay contiguous
        chunk = ALIGN * self.world
        self.total = total
        self.padded = (total + chunk - 1) // chunk * chunk
        self.shard = self.padded // self.world
        self.flat_param = torch.zeros(self.padded, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(self.padded, device=dev, dtype=dt)
        for p, o in zip(params, offs):
            view = self.flat_param[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p._mm_grad_buf = self.flat_grad[o:o + p.numel()].view(p.shape)
            p.grad = None
        self.offsets = offs
        lo = self.rank * self.shard
        self.my_param = self.flat_param[lo:lo + self.shard]
        self.my_grad = self.flat_grad[lo:lo + self.shard]
        self.master = self.my_param.float().clone()
        self.exp_avg = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.shard, device=dev, dtype=torch.float32)
        self._norm_buf = torch.zeros(1, device=dev, dtype=torch.float32)
        self._coef = torch.ones(1, device=dev, dtype=torch.float32)
        self.grad_norm = None

    # ------------------------------------------------------------------ collectives
    def _reduce_scatter_grads(self):
        if self.world == 1:
            return
        backend = dist.get_backend(self.pg)
        if backend == "nccl":           # RCCL
            # bucketed so that each collective moves >= hundreds of MB (launch-amortised on xGMI) while the
            # scratch stays bounded: bucket b covers the same sub-range of every rank's shard.
            # default (bucket_elems == 0): ONE in-place reduce-scatter over the whole flat buffer (RCCL pipelines it
            # internally; output = input + rank*shard is NCCL's documented in-place form)
            per = self.shard if self.bucket_elems <= 0 else max(ALIGN, min(self.shard, self.bucket_elems // self.world // ALIGN * ALIGN))
            for s in range(0, self.shard, per):
                n = min(per, self.shard - s)
                if n == self.shard:
                    dist.reduce_scatter_tensor(self.my_grad, self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
                else:
                    stage = torch.cat([self.flat_grad[r * self.shard + s: r * self.shard + s + n] for r in range(self.world)])
                    dist.reduce_scatter_tensor(self.my_grad[s:s + n], stage, op=dist.ReduceOp.SUM, group=self.pg)
        else:                           # gloo (CPU tests): no reduce_scatter -> all_reduce, keep own shard
            g = self.flat_grad.float() if self.flat_grad.dtype == BF16 else self.flat_grad
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)
            if g is not self.flat_grad:
                self.flat_grad.copy_(g)

    def _all_gather_params(self):
        if self.world == 1:
            return
        backend = dist.get_backend(self.pg)
        if backend == "nccl":
            dist.all_gather_into_tensor(self.flat_param, self.my_param, group=self.pg)
        else:
            parts = [torch.empty_like(self.my_param) for _ in range(self.world)]
            dist.all_gather(parts, self.my_param.clone(), group=self.pg)
            for r, t in enumerate(parts):
                self.flat_param[r * self.shard:(r + 1) * self.shard].copy_(t)

    # ------------------------------------------------------------------ step
    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        # parameters that received no gradient this step contribute zeros
        for p in self.params:
            if p.grad is None:
                p._mm_grad_buf.zero_()
            elif p.grad.data_ptr() != p._mm_grad_buf.data_ptr():
                p._mm_grad_buf.copy_(p.grad)        # a gradient produced outside the flat buffer (e.g. by autograd)
        self._reduce_scatter_grads()
        inv_world = 1.0 / self.world
        # global L2 norm of the MEAN gradient: sqrt(sum over shards) * 1/world
        self._norm_buf.zero_()
        self._sumsq(self.my_grad, self._norm_buf)
        if self.world > 1:
            dist.all_reduce(self._norm_buf, op=dist.ReduceOp.SUM, group=self.pg)
        # coef = min(1, max_norm / (||mean grad|| + 1e-6)) * (1/world), computed on the device (no host sync)
        self._clip_coef_scaled(inv_world)
        self._step += 1
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self._shard_update(self.master, self.exp_avg, self.exp_avg_sq, self.my_grad, self.my_param, float(g["lr"]), b1, b2,
                           g["eps"], g["weight_decay"], self._step, self._coef)
        self._all_gather_params()
        self.grad_norm = self._norm_buf        # sum of squares of the summed gradient (device scalar); see grad_norm_value()
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
