#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP model against the CPU oracle on the tiny golden configuration
(debug aid; needs an MI355X)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from oracle import ref_ops as R
from oracle import ref_model as RM
from oracle.ref_model import OracleConfig, init_state_dict
from metamorph_amd import ops, functional as F
from test_model_gpu import tiny_cfg, hip_model, T

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12)), float((a - b).abs().max())


def main():
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "e2e_mixed_T4_ar1_bf16.npz"))
    cfg = tiny_cfg(num_image_tokens=4)
    sd = init_state_dict(cfg, seed=int(g["seed"]), dtype=torch.bfloat16)
    sdf = {k: v.float() for k, v in sd.items()}                     # fp32 math on the bf16-rounded weights
    model = hip_model(cfg, sd)
    model.train()
    ids, lab, msk, images = T(g["input_ids"]), T(g["labels"]), T(g["attention_mask"]), T(g["images"])
    img_bf = images.bfloat16()
    with torch.no_grad():
        # ---- tower
        raw_ref = RM.siglip_hidden(sdf, cfg, img_bf.float())
        tower = model.get_model().vision_tower
        raw = tower.vision_tower.forward_features(img_bf.to(DEV))
        print("tower raw hidden      rel/max", rel(raw, raw_ref))
        for n_layers in (0, 1):
            r = tower.vision_tower.forward_features(img_bf.to(DEV), select_layer=n_layers)
            c2 = OracleConfig(**{**cfg.__dict__, "v_layers": n_layers})
            print(f"   after {n_layers} layers     ", rel(r, RM.siglip_hidden(sdf, c2, img_bf.float())))
        feat_ref = RM.vision_features(sdf, cfg, img_bf.float())
        feat = tower(img_bf.to(DEV))
        print("tower features        rel/max", rel(feat, feat_ref))
        proj_ref = RM.mm_projector(sdf, cfg, feat_ref)
        proj, tgt = model.encode_images(img_bf.to(DEV))
        print("projected             rel/max", rel(proj, proj_ref))
        # ---- splice
        x_ref, lab_ref, valid, pos_ref, tgt_ref, _ = RM.splice(sdf, cfg, ids, lab, msk, proj_ref, feat_ref)
        out = model.prepare_inputs_labels_for_multimodal(ids.to(DEV), None, msk.to(DEV), None, lab.to(DEV), img_bf.to(DEV))
        emb = out[4]
        print("inputs_embeds         rel/max", rel(emb, x_ref), tuple(emb.shape))
        pd = model._mm_plan[1]
        B, L, h = emb.shape
        # ---- decoder, layer by layer, each fed with the ORACLE's input so errors do not compound
        Hq, Hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        cos, sin = model.model.rope_tables(L, DEV)
        cr, sr = R.rope_tables(torch.arange(L)[None], d, cfg.rope_theta, torch.float32)
        print("rope cos/sin          ", rel(cos[:L], cr[0]), rel(sin[:L], sr[0]))
        meta = F.LayerMeta(B, L, Hq, Hkv, d, cfg.intermediate_size, cfg.rms_norm_eps, cos, sin, pd["seqlens"])
        print("seqlens", pd["seqlens"].tolist(), "valid", valid.sum(1).tolist())
        xin = x_ref.clone()
        for i, layer in enumerate(model.model.layers):
            c1 = OracleConfig(**{**cfg.__dict__, "num_hidden_layers": 1})
            sd1 = {k.replace(f"model.layers.{i}.", "model.layers.0."): v for k, v in sdf.items() if f"model.layers.{i}." in k}
            sd1["model.norm.weight"] = torch.ones(h)
            # oracle single layer without the final norm: reuse llama_decoder pieces
            p = "model.layers.0."
            n = R.rmsnorm(xin, sd1[p + "input_layernorm.weight"], cfg.rms_norm_eps)
            q = R.linear(n, sd1[p + "self_attn.q_proj.weight"]).view(B, L, Hq, d).transpose(1, 2)
            k = R.linear(n, sd1[p + "self_attn.k_proj.weight"]).view(B, L, Hkv, d).transpose(1, 2)
            v = R.linear(n, sd1[p + "self_attn.v_proj.weight"]).view(B, L, Hkv, d).transpose(1, 2)
            cb, sb = R.rope_tables(torch.arange(L)[None].expand(B, L), d, cfg.rope_theta, torch.float32)
            q, k = R.rope_apply(q, cb, sb), R.rope_apply(k, cb, sb)
            a = R.attention(q, k, v, valid, causal=True).transpose(1, 2).reshape(B, L, Hq * d)
            x2 = xin + R.linear(a, sd1[p + "self_attn.o_proj.weight"])
            n2 = R.rmsnorm(x2, sd1[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
            gg = R.linear(n2, sd1[p + "mlp.gate_proj.weight"]); uu = R.linear(n2, sd1[p + "mlp.up_proj.weight"])
            y_ref = x2 + R.linear(R.swiglu(gg, uu), sd1[p + "mlp.down_proj.weight"])
            # HIP pieces
            xd = xin.bfloat16().to(DEV).reshape(B * L, h).contiguous()
            att, mlp = layer.self_attn, layer.mlp
            wqkv = F.fused_weight([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
            n1d = ops.rmsnorm_fwd(xd, layer.input_layernorm.weight, cfg.rms_norm_eps)
            print(f"L{i} rmsnorm1            ", rel(n1d.view(B, L, h), n))
            qkv = ops.gemm(n1d, wqkv)
            qkv_ref = torch.cat([R.linear(n, sd1[p + "self_attn.q_proj.weight"]), R.linear(n, sd1[p + "self_attn.k_proj.weight"]),
                                 R.linear(n, sd1[p + "self_attn.v_proj.weight"])], -1)
            print(f"L{i} qkv gemm            ", rel(qkv.view(B, L, -1), qkv_ref))
            ops.rope_qk_(qkv, B, L, Hq, Hkv, d, cos, sin)
            nq, nk = Hq * d, Hkv * d
            print(f"L{i} q rope              ", rel(qkv[:, :nq].reshape(B, L, Hq, d).transpose(1, 2), q))
            print(f"L{i} k rope              ", rel(qkv[:, nq:nq + nk].reshape(B, L, Hkv, d).transpose(1, 2), k))
            o, lse = ops.attn_fwd(qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:], B, L, Hq, Hkv, d, d ** -0.5, True, pd["seqlens"])
            ov = o.view(B, L, -1)
            for b in range(B):
                nvalid = int(valid[b].sum())
                print(f"L{i} attn sample {b}      ", rel(ov[b, :nvalid], a[b, :nvalid]))
            y, _ = F.decoder_layer_forward(xd, layer, meta)
            yv = y.view(B, L, h)
            print(f"L{i} layer out (valid)   ", rel(yv[valid], y_ref[valid]))
            xin = y_ref
        hid_ref = R.rmsnorm(xin, sdf["model.norm.weight"], cfg.rms_norm_eps)
        hid = ops.rmsnorm_fwd(xin.bfloat16().to(DEV).reshape(B * L, h).contiguous(), model.model.norm.weight, cfg.rms_norm_eps)
        print("final norm            ", rel(hid.view(B, L, h)[valid], hid_ref[valid]))
        logits = ops.gemm(hid, model.lm_head.weight.data, out_f32=True).view(B, L, -1)
        lg_ref = R.linear(hid_ref, sdf["lm_head.weight"])
        print("logits                ", rel(logits[valid], lg_ref[valid]))
    # ---- full model
    out = model(input_ids=ids.to(DEV), attention_mask=msk.to(DEV), labels=lab.to(DEV), images=img_bf.to(DEV))
    ref = RM.forward({k: v.clone().requires_grad_("vision_tower" not in k) for k, v in sdf.items()}, cfg, ids, msk, lab, img_bf.float(), return_logits=False)
    print("loss", float(out.loss.detach()), float(ref["loss"].detach()), "lang", model.loss_language, ref["loss_language"], "img", model.loss_image_ar, ref["loss_image_ar"])
    print("hidden (valid)        ", rel(out.hidden_states[valid], ref["hidden_states"][valid]))


if __name__ == "__main__":
    main()
