"""Per-op CPU restatement (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function states the arithmetic of one kernel row of SURVEY.md section 2.2 in plain
PyTorch, in the dtype conventions of the reference stack (reference call sites are
cited; the arithmetic itself lives in `transformers`, which the reference pins at
4.45.0 in pyproject.toml:16 and which is not vendored under /root/reference).

`dtype` is the "model dtype" (fp32 or bf16).  Where the reference stack upcasts
to fp32 internally (RMSNorm, softmax, interpolation, CE) so does this code.
"""
from __future__ import annotations

import math

import torch

# --------------------------------------------------------------------------------------
# K7  RMSNorm -- transformers LlamaRMSNorm, reached from metamorph_llama.py:349-359
# --------------------------------------------------------------------------------------

def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """fp32 normalise, cast back to the input dtype, THEN multiply by the weight."""
    xf = x.to(torch.float32)
    var = (xf * xf).mean(dim=-1, keepdim=True)
    xn = (xf * torch.rsqrt(var + eps)).to(x.dtype)
    return weight * xn


# --------------------------------------------------------------------------------------
# K9  RoPE (rotate-half convention; cos/sin computed in fp32 then cast to model dtype)
# --------------------------------------------------------------------------------------

def rope_inv_freq(head_dim: int, theta: float, rope_scaling: dict | None = None, max_position_embeddings: int | None = None):
    """Inverse frequencies [d/2] fp32 of the RoPE variant a LLaMA config names -- what HF's LlamaRotaryEmbedding (reached from the reference
    at metamorph_llama.py:349-359 through MetaMorphConfig(LlamaConfig), :129-133) fixes once at construction:
      default   theta^(-2j/d)
      linear    the same divided by `factor`
      llama3    (LLaMA-3.1, the reference README's base model) per wavelength band: longer than ctx/low_freq_factor -> divided by `factor`;
                shorter than ctx/high_freq_factor -> unchanged; between -> linear blend in ctx/wavelength.  ctx =
                original_max_position_embeddings.
    Pinned by tests/golden/r6_rope_tables.npz (HF's own buffers)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    kind = (rope_scaling or {}).get("rope_type") or (rope_scaling or {}).get("type") or "default"
    if kind == "default":
        return inv
    f = float(rope_scaling["factor"])
    if kind == "linear":
        return inv / f
    if kind != "llama3":
        raise NotImplementedError(kind)
    lo, hi = float(rope_scaling["low_freq_factor"]), float(rope_scaling["high_freq_factor"])
    ctx = rope_scaling.get("original_max_position_embeddings") or max_position_embeddings
    lam = 2 * math.pi / inv                                   # wavelength of every band, in positions
    long_band, short_band = lam > ctx / lo, lam < ctx / hi
    stretched = torch.where(long_band, inv / f, inv)
    t = (ctx / lam - lo) / (hi - lo)                          # 0 at the long edge, 1 at the short edge
    blended = (1 - t) * stretched / f + t * stretched
    return torch.where(~short_band & ~long_band, blended, stretched)


def rope_tables(positions: torch.Tensor, head_dim: int, theta: float, dtype: torch.dtype, rope_scaling: dict | None = None,
                max_position_embeddings: int | None = None):
    """positions: int tensor [...]; returns cos, sin of shape [..., head_dim] in `dtype`."""
    inv_freq = rope_inv_freq(head_dim, theta, rope_scaling, max_position_embeddings)     # [d/2]
    ang = positions.to(torch.float32)[..., None] * inv_freq  # [..., d/2]
    emb = torch.cat([ang, ang], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rope_apply(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x: [B, H, L, d]; cos/sin: [B, L, d] (broadcast over heads)."""
    d = x.shape[-1]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    rot = torch.cat([-x2, x1], dim=-1)
    return x * cos[:, None] + rot * sin[:, None]


# --------------------------------------------------------------------------------------
# K10  causal attention with key padding, GQA -- torch SDPA as driven by HF LlamaModel
# --------------------------------------------------------------------------------------

def attention(q, k, v, key_valid=None, causal=True, scale=None):
    """q: [B,Hq,L,d], k/v: [B,Hkv,L,d]; key_valid: bool [B,L] or None.

    Scores and softmax in fp32, output cast to q.dtype.  Query rows that see no key at all
    (cannot happen with right padding + causal) would be NaN exactly like the reference.
    """
    B, Hq, L, d = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    if rep > 1:
        k = k[:, :, None].expand(B, Hkv, rep, L, d).reshape(B, Hq, L, d)
        v = v[:, :, None].expand(B, Hkv, rep, L, d).reshape(B, Hq, L, d)
    scale = d ** -0.5 if scale is None else scale
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    neg = torch.finfo(torch.float32).min
    if causal:
        tri = torch.ones(L, L, dtype=torch.bool).tril()
        s = s.masked_fill(~tri, neg)
    if key_valid is not None:
        s = s.masked_fill(~key_valid[:, None, None, :], neg)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v.float()).to(q.dtype)


# --------------------------------------------------------------------------------------
# K12  SwiGLU,  K2/K5/K15 GELU variants,  K2 LayerNorm
# --------------------------------------------------------------------------------------

def swiglu(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.silu(gate) * up


def gelu_erf(x):  # nn.GELU() default: mm_projector (multimodal_projector/builder.py:57), vision_head (metamorph_llama.py:254)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def gelu_tanh(x):  # SigLIP hidden_act "gelu_pytorch_tanh"
    c = math.sqrt(2.0 / math.pi)
    return 0.5 * x * (1.0 + torch.tanh(c * (x + 0.044715 * x * x * x)))


def layernorm(x, weight, bias, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def linear(x, w, b=None):
    return torch.nn.functional.linear(x, w, b)


# --------------------------------------------------------------------------------------
# K3 + K4  729 -> T token reduction (fp32 bilinear, align_corners=False) and L2 normalise
#          siglip_encoder.py:151-163 and :206-208
# --------------------------------------------------------------------------------------

def _lerp_index(out_size: int, in_size: int):
    """Source indices / weights of torch's bilinear kernel with align_corners=False."""
    scale = in_size / out_size
    i0, i1, w1 = [], [], []
    for o in range(out_size):
        src = scale * (o + 0.5) - 0.5
        if src < 0.0:
            src = 0.0
        a = int(src)
        b = a + (1 if a < in_size - 1 else 0)
        i0.append(a)
        i1.append(b)
        w1.append(src - a)
    return i0, i1, w1


def bilinear_reduce(feat: torch.Tensor, num_tokens: int) -> torch.Tensor:
    """feat: [N, P, C] with P a square; returns [N, num_tokens, C] in feat.dtype (fp32 math)."""
    N, P, C = feat.shape
    if P == num_tokens:
        return feat
    side_in = int(math.isqrt(P))
    side_out = int(math.isqrt(num_tokens))
    g = feat.to(torch.float32).view(N, side_in, side_in, C)
    y0, y1, wy = _lerp_index(side_out, side_in)
    x0, x1, wx = _lerp_index(side_out, side_in)
    out = torch.empty(N, side_out, side_out, C, dtype=torch.float32)
    for oy in range(side_out):
        for ox in range(side_out):
            ly, lx = wy[oy], wx[ox]
            top = g[:, y0[oy], x0[ox]] * (1.0 - lx) + g[:, y0[oy], x1[ox]] * lx
            bot = g[:, y1[oy], x0[ox]] * (1.0 - lx) + g[:, y1[oy], x1[ox]] * lx
            out[:, oy, ox] = top * (1.0 - ly) + bot * ly
    return out.to(feat.dtype).reshape(N, side_out * side_out, C)


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    n = x.norm(p=2, dim=-1, keepdim=True).clamp_min(eps)
    return x / n


# --------------------------------------------------------------------------------------
# K13  shifted cross-entropy (metamorph_llama.py:398-413)
# --------------------------------------------------------------------------------------

def shifted_cross_entropy(logits_f32: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100):
    """logits_f32: [B,L,V] float32; labels [B,L]. mean over labels[:,1:] != ignore_index."""
    V = logits_f32.shape[-1]
    lg = logits_f32[:, :-1].reshape(-1, V)
    lb = labels[:, 1:].reshape(-1)
    keep = lb != ignore_index
    lse = torch.logsumexp(lg, dim=-1)
    picked = lg.gather(1, lb.clamp_min(0)[:, None])[:, 0]
    nll = (lse - picked)[keep]
    return nll.sum() / keep.sum()          # 0/0 -> NaN exactly like nn.CrossEntropyLoss


# --------------------------------------------------------------------------------------
# K16  image-AR losses (metamorph_llama.py:434-459, 211-219)
# --------------------------------------------------------------------------------------

def cosine_loss(target: torch.Tensor, pred: torch.Tensor, eps: float = 1e-8):
    """-mean_r cos(target_r, pred_r) with F.cosine_similarity's eps clamp."""
    tf, pf = target, pred
    dot = (tf * pf).sum(-1)
    nt = tf.norm(dim=-1).clamp_min(eps)
    np_ = pf.norm(dim=-1).clamp_min(eps)
    return -(dot / (nt * np_)).mean()


def soft_ce_loss(target: torch.Tensor, pred_prob: torch.Tensor):
    return -(target * torch.log(pred_prob + 1e-10)).sum(dim=1).mean()


def mean_abs_loss(target: torch.Tensor, pred: torch.Tensor):
    """The reference's `mse_loss_fn(z=target, h=pred)` (metamorph_llama.py:211-219): walks zip(target rows, pred rows), adds the mean
    |z_i - h_i| of each pair and divides by len(target) -- with equal row counts the mean |z - h|; with unequal counts the first
    min(Rt, R) pairs over Rt (pinned by tests/golden/ops_r3.npz)."""
    n = min(target.shape[0], pred.shape[0])
    return (target[:n] - pred[:n]).abs().mean(dim=-1).sum() / target.shape[0]


# --------------------------------------------------------------------------------------
# ZeRO-2 shard update: AdamW exactly as torch.optim.AdamW (HF optim="adamw_torch", train.py:82)
# --------------------------------------------------------------------------------------

def adamw_step(p32, g, m, v, step: int, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, grad_scale: float = 1.0):
    """In-place on fp32 master p32 and moments m, v; g may be bf16/fp32.  Returns p32."""
    gf = g.to(torch.float32) * grad_scale
    p32.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(gf, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(gf, gf, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p32.addcdiv_(m, denom, value=-lr / bc1)
    return p32
