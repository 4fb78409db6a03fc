"""SigLIP vision tower on libmm355 kernels (forward only; the tower is frozen in every shipped recipe).

Mirrors the reference's `SiglipVisionTower` (reference metamorph/model/multimodal_encoder/siglip_encoder.py:62-237):
same constructor arguments, attributes and `forward(images) -> [N, num_image_tokens, 1152]` contract --
hidden_states[select_layer = -1] of the HF SigLIP encoder (i.e. the last encoder layer's output BEFORE
post_layernorm), 729 -> T tokens by fp32 bilinear interpolation, optional L2 normalisation.

What differs is only the machinery: patch embedding is an im2col + MFMA GEMM with the bias and position
embedding fused in the epilogue, q/k/v are one fused GEMM, attention is the non-causal flash kernel
(d = 72 padded to 96 in LDS), fc1 carries bias + tanh-GELU in its epilogue, and the post_layernorm /
attention-pool head that the reference computes and discards are simply not computed.
"""
from __future__ import annotations

import re

import math

import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ..modules import HipLayerNorm, HipLinear

BF16 = torch.bfloat16

SO400M_14_384 = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                     image_size=384, patch_size=14, layer_norm_eps=1e-6)


_TOWER_NAME = re.compile(r"(siglip/CLIP-ViT-SO400M-14|timm/ViT-SO400M-14-SigLIP)(-384)?")
_NAME_FIELD = re.compile(r"(res|interp)(.*)")


def extract_res_interp(model_name):
    """`--vision_tower` name grammar of the reference (siglip_encoder.py:34-59): one of two spellings of SigLIP-SO400M/14, optionally
    the 384-pixel checkpoint, then `-res<N>` (input resolution; default 384 if "384" occurs anywhere in the name, else 224) and
    `-interp<N>` (token count after interpolation) in any order.  Returns (timm hub name, resolution, interp or None); unknown towers and
    malformed fields raise ValueError."""
    m = _TOWER_NAME.match(model_name)
    if m is None:
        raise ValueError(f"Unknown vision tower: {model_name}")
    fields = {"res": 384 if "384" in model_name else 224, "interp": None}
    for token in model_name.split("-"):
        f = _NAME_FIELD.fullmatch(token)
        if f:
            fields[f.group(1)] = int(f.group(2))           # "-res" / "-resnet": ValueError, as in the reference
    return "hf-hub:timm/ViT-SO400M-14-SigLIP" + (m.group(2) or ""), fields["res"], fields["interp"]


class _PatchEmbedding(nn.Module):
    def __init__(self, hv, p):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(hv, 3, p, p))
        self.bias = nn.Parameter(torch.zeros(hv))
        nn.init.normal_(self.weight, std=0.02)


class _PosEmbedding(nn.Module):
    def __init__(self, n, hv):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, hv))
        nn.init.normal_(self.weight, std=0.02)


class _Embeddings(nn.Module):
    def __init__(self, g):
        super().__init__()
        self.patch_embedding = _PatchEmbedding(g["hidden_size"], g["patch_size"])
        self.position_embedding = _PosEmbedding((g["image_size"] // g["patch_size"]) ** 2, g["hidden_size"])


class _Attention(nn.Module):
    def __init__(self, hv):
        super().__init__()
        self.k_proj = HipLinear(hv, hv)
        self.v_proj = HipLinear(hv, hv)
        self.q_proj = HipLinear(hv, hv)
        self.out_proj = HipLinear(hv, hv)


class _MLP(nn.Module):
    def __init__(self, hv, iv):
        super().__init__()
        self.fc1 = HipLinear(hv, iv)
        self.fc2 = HipLinear(iv, hv)


class _EncoderLayer(nn.Module):
    def __init__(self, g):
        super().__init__()
        hv = g["hidden_size"]
        self.layer_norm1 = HipLayerNorm(hv, g["layer_norm_eps"])
        self.self_attn = _Attention(hv)
        self.layer_norm2 = HipLayerNorm(hv, g["layer_norm_eps"])
        self.mlp = _MLP(hv, g["intermediate_size"])


class _Encoder(nn.Module):
    def __init__(self, g):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(g) for _ in range(g["num_hidden_layers"])])


class _Cfg:
    def __init__(self, g):
        self.__dict__.update(g)


class HipSiglipVisionTransformer(nn.Module):
    """State-dict layout of HF `SiglipVisionTransformer` (what the reference stores as `vision_tower.vision_tower`)."""

    def __init__(self, geometry=None):
        super().__init__()
        g = dict(SO400M_14_384)
        g.update(geometry or {})
        self.geometry = g
        self.config = _Cfg(g)
        self.embeddings = _Embeddings(g)
        self.encoder = _Encoder(g)
        self.post_layernorm = HipLayerNorm(g["hidden_size"], g["layer_norm_eps"])   # kept for checkpoint round trips; unused
        self._cache = {}
        self.gradient_checkpointing = False       # set by HF gradient_checkpointing_enable(); used when the tower trains

    @property
    def dtype(self):
        return self.embeddings.patch_embedding.weight.dtype

    @property
    def device(self):
        return self.embeddings.patch_embedding.weight.device

    # -- derived, cached operands (the tower is frozen, so they are rebuilt only when storage moves) --------
    def _cached(self, key, srcs, build):
        from ...functional import param_generation
        # (pointer, torch version) catches re-loads and in-place torch updates; the generation counter catches optimizer steps that
        # write parameters through raw pointers (a trainable tower under Zero2AdamW: eval / generation after training steps)
        gen = param_generation() if any(s.requires_grad for s in srcs) else 0      # a frozen tower is never touched by the optimizer
        sig = (gen,) + tuple((s.data_ptr(), s._version) for s in srcs)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            hit = (sig, build())
            self._cache[key] = hit
        return hit[1]

    def _patch_weight(self):
        w = self.embeddings.patch_embedding.weight
        k = w.shape[1] * w.shape[2] * w.shape[3]
        kp = (k + 7) // 8 * 8

        def build():
            buf = torch.zeros((w.shape[0], kp), device=w.device, dtype=BF16)
            buf[:, :k].copy_(w.data.reshape(w.shape[0], k))
            return buf
        return self._cached("patch", [w], build), kp

    def _qkv(self, j):
        a = self.encoder.layers[j].self_attn
        ws = [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight]
        bs = [a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]
        return self._cached(("qkv", j), ws + bs, lambda: (torch.cat([w.data for w in ws], 0).contiguous(),
                                                         torch.cat([b.data for b in bs], 0).contiguous()))

    def _mlp_padded(self, j):
        """fc1 / fc2 operands of layer j with the intermediate width rounded up to whole 128-wide K tiles (SO400M: 4304 -> 4352) so
        that both GEMMs run on the LDS-DMA kernels instead of the register-staged fallback for ragged K: fc1 gains zero rows and
        zero bias entries (tanh-GELU(0) = 0), fc2 zero columns -- the product is unchanged bit for bit (adding exact zeros)."""
        m = self.encoder.layers[j].mlp
        iv = m.fc1.weight.shape[0]
        ivp = (iv + 127) // 128 * 128
        if ivp == iv:
            return m.fc1.weight.data, m.fc1.bias.data, m.fc2.weight.data

        def build():
            w1 = torch.zeros((ivp, m.fc1.weight.shape[1]), device=m.fc1.weight.device, dtype=BF16)
            w1[:iv].copy_(m.fc1.weight.data)
            b1 = torch.zeros((ivp,), device=w1.device, dtype=BF16)
            b1[:iv].copy_(m.fc1.bias.data)
            w2 = torch.zeros((m.fc2.weight.shape[0], ivp), device=w1.device, dtype=BF16)
            w2[:, :iv].copy_(m.fc2.weight.data)
            return w1, b1, w2
        return self._cached(("mlp", j), [m.fc1.weight, m.fc1.bias, m.fc2.weight], build)

    @torch.no_grad()
    def forward_features(self, images, select_layer=-1):
        """[N,3,H,W] (fp32 or bf16) -> hidden_states[select_layer] of the HF encoder, [N, P, hv] bf16
        (index -1 = output of the last encoder layer, before post_layernorm; 0 = the embeddings)."""
        g = self.geometry
        if self.dtype != BF16:
            raise TypeError("SigLIP tower: the MI355X kernels compute in bf16; call .to(torch.bfloat16)")
        if images.dtype not in (torch.float32, BF16):
            images = images.float()
        images = images.contiguous()
        N, _, H, W = images.shape
        p, hv, heads = g["patch_size"], g["hidden_size"], g["num_attention_heads"]
        d = hv // heads
        P = (H // p) * (W // p)
        pos = self.embeddings.position_embedding.weight.data
        if P != pos.shape[0]:
            raise NotImplementedError("position-embedding interpolation for non-native resolutions is not implemented")
        wpe, kp = self._patch_weight()
        cols = ops.im2col_patch(images, p, kp)
        x = ops.gemm(cols, wpe, bias=self.embeddings.patch_embedding.bias.data, residual=pos, res_row_mod=P)
        del cols
        n_layers = len(self.encoder.layers)
        run = select_layer if select_layer >= 0 else n_layers + 1 + select_layer
        if not 0 <= run <= n_layers:
            raise IndexError(f"mm_vision_select_layer={select_layer} out of range for {n_layers} layers")
        for j, layer in enumerate(self.encoder.layers[:run]):
            h1 = layer.layer_norm1(x)
            wqkv, bqkv = self._qkv(j)
            qkv = ops.gemm(h1, wqkv, bias=bqkv)
            o, _ = ops.attn_fwd(qkv[:, :hv], qkv[:, hv:2 * hv], qkv[:, 2 * hv:], N, P, heads, heads, d, d ** -0.5, False, None)
            a = layer.self_attn
            x = ops.gemm(o, a.out_proj.weight.data, bias=a.out_proj.bias.data, residual=x)
            h2 = layer.layer_norm2(x)
            w1, b1, w2 = self._mlp_padded(j)
            f = ops.gemm(h2, w1, bias=b1, gelu="tanh")
            x = ops.gemm(f, w2, bias=layer.mlp.fc2.bias.data, residual=x)
        return x.view(N, P, hv)

    def forward_features_train(self, images, select_layer=-1):
        """The same computation with a backward pass (freeze_vision=False; reference siglip_encoder.py:138-141): autograd nodes
        PatchEmbedFn / SiglipLayerFn write the tower's parameter gradients straight into their gradient buffers."""
        from ... import functional as F
        g = self.geometry
        if self.dtype != BF16:
            raise TypeError("SigLIP tower: the MI355X kernels compute in bf16; call .to(torch.bfloat16)")
        if images.dtype not in (torch.float32, BF16):
            images = images.float()
        images = images.contiguous()
        N, _, H, W = images.shape
        p, hv, heads = g["patch_size"], g["hidden_size"], g["num_attention_heads"]
        P = (H // p) * (W // p)
        if P != self.embeddings.position_embedding.weight.shape[0]:
            raise NotImplementedError("position-embedding interpolation for non-native resolutions is not implemented")
        n_layers = len(self.encoder.layers)
        run = select_layer if select_layer >= 0 else n_layers + 1 + select_layer
        if not 0 <= run <= n_layers:
            raise IndexError(f"mm_vision_select_layer={select_layer} out of range for {n_layers} layers")
        x = F.PatchEmbedFn.apply(images, self.embeddings, *self.embeddings.parameters())
        geo = F.SiglipGeo(N, P, heads, hv // heads, g["layer_norm_eps"], recompute=bool(getattr(self, "gradient_checkpointing", False)))
        for layer in self.encoder.layers[:run]:
            x = F.SiglipLayerFn.apply(x, layer, geo, *layer.parameters())
        return x.view(N, P, hv)


class SiglipVisionTower(nn.Module):
    def __init__(self, vision_tower_name, args, delay_load=False):
        super().__init__()
        base_model_name, res, interp = extract_res_interp(vision_tower_name)
        self.is_loaded = False
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self.image_token_reduction = getattr(args, "image_token_reduction", "none")
        self.image_token_len = getattr(args, "num_image_tokens", 256)
        self.freeze_vision = getattr(args, "freeze_vision", False)
        self.vision_coef = getattr(args, "vision_coef", 1.0)
        self.normalize_vision = getattr(args, "normalize_vision", False)
        self.apply_softmax = getattr(args, "apply_softmax", False)
        self.geometry = dict(SO400M_14_384)
        self.geometry.update(getattr(args, "mm_vision_geometry", None) or {})
        self.vision_tower_name = base_model_name
        self._image_size = res if res is not None else 512
        self._interp_size = interp
        self.hidden_size = self.geometry["hidden_size"]
        self.image_processor = None
        if self.image_token_reduction == "concat_interpolation":     # reference siglip_encoder.py:109-110
            self.hidden_size = 4 * self.geometry["hidden_size"]
        if not delay_load:
            self.load_model()

    def load_model(self, device_map=None, state_dict=None, random_init=False):
        """Builds the tower.  Weights come from `state_dict` (HF SigLIP vision keys), from the HF hub checkpoint
        the reference hard-codes (google/siglip-so400m-patch14-384, siglip_encoder.py:113) when reachable, or are
        random (benchmarks / tests)."""
        self.vision_model = "siglip"
        self.vision_tower = HipSiglipVisionTransformer(self.geometry)
        if state_dict is None and not random_init:
            from transformers import AutoModel, AutoProcessor          # network / cache access, like the reference
            model = AutoModel.from_pretrained("google/siglip-so400m-patch14-384")
            self.image_processor = AutoProcessor.from_pretrained("google/siglip-so400m-patch14-384").image_processor
            self.image_processor.crop_size = {"height": 384, "width": 384}
            state_dict = {k[len("vision_model."):]: v for k, v in model.state_dict().items() if k.startswith("vision_model.")}
        if self.image_processor is None:                             # offline: the same checkpoint's preprocessor constants
            from ...image_processing import SiglipImageProcessor
            self.image_processor = SiglipImageProcessor(size=384)
        if state_dict is not None:
            own = self.vision_tower.state_dict()
            self.vision_tower.load_state_dict({k: v for k, v in state_dict.items() if k in own}, strict=False)
        hv = self.geometry["hidden_size"]
        self.hidden_size = 4 * hv if self.image_token_reduction == "concat_interpolation" else hv
        if self.image_token_reduction == "mlpmixer" and not hasattr(self, "token_mixer"):     # reference siglip_encoder.py:100-107
            patches = (self.geometry["image_size"] // self.geometry["patch_size"]) ** 2
            self.token_mixer = nn.Sequential(HipLinear(patches, self.image_token_len))
            self.channel_mixer = nn.Sequential(HipLinear(hv, hv))
            if state_dict is not None:
                for name, mod in (("token_mixer", self.token_mixer), ("channel_mixer", self.channel_mixer)):
                    sub = {k[len(name) + 1:]: v for k, v in state_dict.items() if k.startswith(name + ".")}
                    if sub:
                        mod.load_state_dict(sub)
        self.is_loaded = True

    def feature_select(self, hidden_last):
        if self.select_feature not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        return hidden_last

    def forward(self, images):
        train = not self.freeze_vision and torch.is_grad_enabled() and any(p.requires_grad for p in self.vision_tower.parameters())
        if train:                                                # reference: torch.set_grad_enabled(not self.freeze_vision)
            feats = self.feature_select(self.vision_tower.forward_features_train(images, self.select_layer))
        else:
            feats = self.feature_select(self.vision_tower.forward_features(images, self.select_layer))
        b, num_tokens, dim = feats.shape
        side_in = int(math.isqrt(num_tokens))
        side_out = side_in
        from ... import functional as F
        reduction = "interpolation"
        if num_tokens != self.image_token_len:
            if self.image_token_len == -1:
                return torch.zeros((b, num_tokens, dim), device=feats.device, dtype=feats.dtype)
            reduction = self.image_token_reduction
            if reduction == "interpolation":
                side_out = int(np.random.randint(1, 25)) if self.image_token_len == 0 else int(np.sqrt(self.image_token_len))
            elif reduction == "mlpmixer":                        # token mixing over the patch axis, then a channel Linear (:164-168)
                feats = self._mlpmixer(feats, train)
            elif reduction == "concat_interpolation":            # interpolate to 4 T tokens, concatenate every 2 x 2 block (:169-199)
                feats = self._concat_interpolation(feats, side_in, train)
                side_in = side_out = int(math.isqrt(self.image_token_len))
            else:
                raise NotImplementedError("Not Implemented!")
        if reduction != "interpolation":
            side_in = side_out = 1                               # nothing left to interpolate: only the optional L2 normalisation below
            b, t, c = feats.shape
            feats = feats.reshape(b * t, 1, c)
        if side_out != side_in or self.normalize_vision:
            if train:
                feats = F.BilinearL2NormFn.apply(feats.contiguous(), side_in, side_out, bool(self.normalize_vision))
            else:
                feats = ops.bilinear_l2norm(feats.contiguous(), side_in, side_out, bool(self.normalize_vision))
        if reduction != "interpolation":
            feats = feats.reshape(b, t, c)
        if self.apply_softmax:                                   # softmax(feat / 0.07) (siglip_encoder.py:210-211)
            n, t, c = feats.shape
            flat = feats.reshape(n * t, c).contiguous()
            flat = F.SoftmaxRowsFn.apply(flat, 0.07) if train else ops.softmax_rows(flat, 0.07)
            feats = flat.view(n, t, c)
        return feats

    def _mlpmixer(self, feats, train):
        """[b, P, c] -> token_mixer over P (Linear(P, T) applied to the transposed features) -> [b, T, c] -> channel_mixer.
        The contraction runs over the patch axis, so each image is transposed ([c, P], zero-padded to a legal GEMM K) first."""
        from ... import functional as F
        b, P, c = feats.shape
        tm, cm = self.token_mixer[0], self.channel_mixer[0]
        T = tm.weight.shape[0]
        Pp, Tp = (P + 7) // 8 * 8, (T + 7) // 8 * 8              # legal GEMM K / leading dimensions: zero padding, sliced off below
        wt, bt = tm.weight, tm.bias
        if Pp != P or Tp != T:
            wt, bt = F.PadFn.apply(tm.weight, Tp, Pp), F.PadFn.apply(tm.bias, Tp, None)
        outs = []
        for i in range(b):
            xt = F.TransposeFn.apply(feats[i].contiguous(), Pp)  # [c, Pp]
            y = F.LinearWBFn.apply(xt, wt, bt)                   # [c, Tp]  (bias along the token axis)
            outs.append(F.TransposeFn.apply(y, None)[:T])        # [T, c]
        y = (torch.cat(outs, 0) if b > 1 else outs[0]).contiguous()
        return cm(y).view(b, T, c)

    def _concat_interpolation(self, feats, side_in, train):
        from ... import functional as F
        b, _, c = feats.shape
        s = int(math.isqrt(self.image_token_len))
        mid = 2 * s                                              # intermediate grid: 4 T tokens
        if mid != side_in:
            x = (F.BilinearL2NormFn.apply(feats.contiguous(), side_in, mid, False) if train
                 else ops.bilinear_l2norm(feats.contiguous(), side_in, mid, False))
        else:
            x = feats.contiguous()
        # out[b, (I, J), (di * 2 + dj) * c + ch] = x[b, (2 I + di) * mid + 2 J + dj, ch]: one row gather
        I, J, di, dj = np.meshgrid(np.arange(s), np.arange(s), np.arange(2), np.arange(2), indexing="ij")
        src = ((2 * I + di) * mid + 2 * J + dj).reshape(-1)
        idx = (np.arange(b)[:, None] * (mid * mid) + src[None, :]).reshape(-1).astype(np.int32)
        idx_d = torch.from_numpy(idx).to(feats.device)
        x2 = x.reshape(b * mid * mid, c)
        g = F.RowsGatherFn.apply(x2, idx_d) if (train and x2.requires_grad) else ops.rows_gather(x2, idx_d)
        return g.view(b, s * s, 4 * c)

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def config(self):
        return self.vision_tower.config if self.is_loaded else _Cfg(self.geometry)

    @property
    def num_patches_per_side(self):
        return self.geometry["image_size"] // self.geometry["patch_size"]
