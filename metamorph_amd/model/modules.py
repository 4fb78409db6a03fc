"""Parameter containers whose forward is a libmm355 kernel (never a torch op).

They keep the attribute names of the torch modules the reference uses (weight / bias), so state-dict keys
are unchanged, but calling them on a CPU tensor raises instead of silently computing with ATen.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as F
from .. import ops

BF16 = torch.bfloat16


def _require_bf16(t, what):
    if t.dtype != BF16:
        raise TypeError(f"{what}: the MI355X kernels compute in bf16 (fp32 accumulate); got {t.dtype}. "
                        "Convert the model with .to(torch.bfloat16).")


class HipLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.weight, std=0.02)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x):
        _require_bf16(self.weight, "HipLinear")
        shp = x.shape
        y = F.linear(x.reshape(-1, shp[-1]).contiguous(), self)
        return y.view(*shp[:-1], self.out_features)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class HipGELU(nn.Module):
    def __init__(self, approximate="none"):
        super().__init__()
        self.kind = ops.GELU_TANH if approximate == "tanh" else ops.GELU_ERF

    def forward(self, x):
        return F.GeluFn.apply(x.contiguous(), self.kind)


class HipRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        _require_bf16(self.weight, "HipRMSNorm")
        return F.RmsNormFn.apply(x.contiguous(), self.weight, self.variance_epsilon)


class HipLayerNorm(nn.Module):
    """Forward only (the SigLIP tower is frozen in every recipe of the reference)."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.eps = eps

    def forward(self, x):
        return ops.layernorm_fwd(x.contiguous(), self.weight.data, self.bias.data, self.eps)


class HipEmbedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight, std=0.02)

    def forward(self, ids):
        """Plain lookup.  Multimodal training goes through the splice plan; a direct call under autograd (text-only batches,
        forward(images=None)) gets its gradient from `EmbeddingFn`."""
        _require_bf16(self.weight, "HipEmbedding")
        if torch.is_grad_enabled() and self.weight.requires_grad and ids.numel() > 0:
            return F.EmbeddingFn.apply(self.weight, self, ids)
        flat = ids.reshape(-1).to(torch.int32)
        out = ops.splice_gather(self.weight.data, None, flat, self.embedding_dim)
        return out.view(*ids.shape, self.embedding_dim)


class HipSoftmax(nn.Module):
    """nn.Softmax(dim=-1) over the feature dimension (the 'mlpsoftmax' connector, reference multimodal_projector/builder.py:45-50);
    `temperature` divides the input first (encode_images(return_prob=True): softmax(x / temperature_in), metamorph_arch.py:151)."""

    def __init__(self, dim=-1):
        super().__init__()
        if dim != -1:
            raise NotImplementedError("HipSoftmax: only the feature dimension (dim=-1) has a kernel")

    def forward(self, x, temperature=1.0):
        shp = x.shape
        y = F.SoftmaxRowsFn.apply(x.reshape(-1, shp[-1]).contiguous(), float(temperature))
        return y.view(shp)
