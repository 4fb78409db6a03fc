// C ABI of the attention kernels (attn2.hip) plus the two small helper kernels of the backward pass:
//   attn_delta_kernel    delta[b][h][l] = sum_d dO * O            (the softmax-backward row term)
//   group_reduce_kernel  sums the per-query-head fp32 dK/dV partials of a GQA group into the bf16 column block
// Replaces torch SDPA as driven by HF LlamaModel (causal + key padding, GQA; reference call site
// metamorph_llama.py:349-359) and by HF SiglipAttention (non-causal, d = 72; siglip_encoder.py:141).
#include "mm355_common.h"
#include "attn2.h"
#include <cstdlib>
#include <algorithm>

namespace {

constexpr int NT = 256;

// 4 lanes per row: delta
__global__ __launch_bounds__(NT) void attn_delta_kernel(const uint16_t* __restrict__ o, const uint16_t* __restrict__ d_o, int64_t ld_o,
                                                        float* __restrict__ delta, int L, int Hq, int d) {
    const int l = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const int h = blockIdx.y, b = blockIdx.z;
    const int dv = d >> 3;
    float s = 0.f;
    if (l < L) {
        for (int v = part; v < dv; v += 4) {
            float x[8], y[8];
            unpack8(*(const u32x4*)(o + ((int64_t)b * L + l) * ld_o + (int64_t)h * d + v * 8), x);
            unpack8(*(const u32x4*)(d_o + ((int64_t)b * L + l) * ld_o + (int64_t)h * d + v * 8), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[e] * y[e];
        }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0 && l < L) delta[((int64_t)b * Hq + h) * L + l] = s;
}

// dk[m][hk*d + c] = sum over the GQA group of the per-query-head fp32 partials
__global__ __launch_bounds__(NT) void group_reduce_kernel(const float* __restrict__ part, uint16_t* __restrict__ out, int64_t ld_out, int64_t rows,
                                                          int Hq, int Hkv, int d) {
    const int group = Hq / Hkv, cv = (Hkv * d) >> 3;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t r = i / cv; const int c = (int)(i % cv) * 8;
        const int hk = c / d, cc = c % d;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < group; ++g) {
            const float* p = part + r * ((int64_t)Hq * d) + (int64_t)(hk * group + g) * d + cc;
            const f32x4 x0 = *(const f32x4*)p, x1 = *(const f32x4*)(p + 4);
            acc[0] += x0.x; acc[1] += x0.y; acc[2] += x0.z; acc[3] += x0.w;
            acc[4] += x1.x; acc[5] += x1.y; acc[6] += x1.z; acc[7] += x1.w;
        }
        *(u32x4*)(out + r * ld_out + c) = pack8(acc);
    }
}

int pick_dp(int64_t d) { return d <= 64 ? 64 : (d <= 96 ? 96 : 128); }

bool bad_geom(int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d) {
    return B <= 0 || L <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || d <= 0 || d > 128 || (d & 7) || L > 0x7fffffff;
}

// d == 128 goes to the LDS-DMA kernels (the generic ones stay reachable through the *_variant entries, variant 2: A/B and tests)
bool fast128(int64_t d, int64_t ld_k) { return d == 128 && ld_k * 2 * 64 < 0x7fffffff; }

}  // namespace

namespace {
// variant: 0 = the product's choice, 2 = the generic-d kernels of attn2.hip, 3 = the two-waves-per-SIMD kernels of attn3.hip (round 2;
// compiled only with -DMM355_LEGACY_VARIANTS, MM355_EUNSUPPORTED otherwise), 4 = the one-wave-per-SIMD stream of attn4.hip, 41 = attn4's
// serialised debugging stream (bit-identical to 4 by construction; tests / tools only)
int attn_fwd_impl(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, mm355_bf16* o,
                  int64_t ld_o, float* lse, const int32_t* seqlens, int64_t B, int64_t L, int64_t Hq, int64_t Hkv,
                  int64_t d, float scale, int causal, int variant, void* stream, int32_t* dbg = nullptr) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!q || !k || !v || !o || !lse || bad_geom(B, L, Hq, Hkv, d)) return MM355_EINVAL;
    if ((ld_q & 7) || (ld_k & 7) || (ld_o & 7)) return MM355_EINVAL;
    attn2::Args a{q, k, v, nullptr, ld_q, ld_k, ld_o, o, lse, nullptr, nullptr, nullptr, 0, seqlens,
                  nullptr, nullptr, nullptr, nullptr, 0, (int)B, (int)L, (int)Hq, (int)Hkv, (int)d, scale, causal, nullptr, nullptr, nullptr, dbg};
    const bool can4 = d == 128 && L * ld_k * 2 < 0x7fffffffll;          // 32-bit descriptor offsets inside one sample
    if (variant == 4 || variant == 41) return can4 ? mm355_attn4_fwd_launch(a, variant == 41, (hipStream_t)stream) : MM355_EUNSUPPORTED;
#ifdef MM355_LEGACY_VARIANTS                                 // tools build (MM355_LEGACY_VARIANTS=1 python -m metamorph_amd.build): round-2 kernels for A/B timing
    if (variant == 3) return fast128(d, ld_k) ? mm355_attn3_fwd_launch(a, (hipStream_t)stream) : MM355_EUNSUPPORTED;
#else
    if (variant == 3) return MM355_EUNSUPPORTED;
#endif
    if (variant == 2) return mm355_attn2_fwd_launch(a, pick_dp(d), (hipStream_t)stream);
    if (variant != 0) return MM355_EINVAL;
    if (can4 && fast128(d, ld_k)) return mm355_attn4_fwd_launch(a, 0, (hipStream_t)stream);
    return mm355_attn2_fwd_launch(a, pick_dp(d), (hipStream_t)stream);   // generic head sizes (SigLIP d = 72, TinyLlama d = 64), samples of >= 2 GiB
}
}  // namespace

extern "C" int mm355_attn_fwd(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, mm355_bf16* o,
                              int64_t ld_o, float* lse, const int32_t* seqlens, int64_t B, int64_t L, int64_t Hq, int64_t Hkv,
                              int64_t d, float scale, int causal, void* stream) {
    return attn_fwd_impl(q, k, v, ld_q, ld_k, o, ld_o, lse, seqlens, B, L, Hq, Hkv, d, scale, causal, 0, stream);
}

extern "C" int mm355_attn_fwd_variant(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, mm355_bf16* o,
                                      int64_t ld_o, float* lse, const int32_t* seqlens, int64_t B, int64_t L, int64_t Hq, int64_t Hkv,
                                      int64_t d, float scale, int causal, int variant, void* stream) {
    return attn_fwd_impl(q, k, v, ld_q, ld_k, o, ld_o, lse, seqlens, B, L, Hq, Hkv, d, scale, causal, variant, stream);
}

extern "C" int mm355_attn_fwd_debug(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, mm355_bf16* o,
                                    int64_t ld_o, float* lse, const int32_t* seqlens, int64_t B, int64_t L, int64_t Hq, int64_t Hkv,
                                    int64_t d, float scale, int causal, int variant, int32_t* rescale_counts, void* stream) {
    if (!rescale_counts || (variant != 4 && variant != 41)) return MM355_EINVAL;
    return attn_fwd_impl(q, k, v, ld_q, ld_k, o, ld_o, lse, seqlens, B, L, Hq, Hkv, d, scale, causal, variant, stream, rescale_counts);
}

extern "C" int mm355_attn_bwd_prep(const mm355_bf16* o, const mm355_bf16* d_o, int64_t ld_o, float* delta, int64_t B, int64_t L,
                                   int64_t Hq, int64_t d, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!o || !d_o || !delta || B <= 0 || L <= 0 || Hq <= 0 || d <= 0 || d > 128 || (d & 7) || (ld_o & 7)) return MM355_EINVAL;
    dim3 grid((unsigned)((L + 63) / 64), (unsigned)Hq, (unsigned)B);
    hipLaunchKernelGGL(attn_delta_kernel, grid, dim3(NT), 0, (hipStream_t)stream, o, d_o, ld_o, delta, (int)L, (int)Hq, (int)d);
    return mm_launch_status();
}

extern "C" int64_t mm355_attn_bwd_ws_floats(int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, int64_t ld_max) {
    if (fast128(d, ld_max)) return 2 * B * Hq * L;           // d == 128: -lse * log2(e) and -delta, the C operands of the score chains
    if (Hq == Hkv) return 0;                                 // no GQA
    return 2 * B * L * Hq * d;
}

namespace {
int attn_bwd_impl(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, const mm355_bf16* d_o,
                  int64_t ld_o, const float* lse, const float* delta, const int32_t* seqlens, mm355_bf16* dq, int64_t ld_dq,
                  mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d,
                  float scale, int causal, float* workspace, const uint16_t* rope_cos, const uint16_t* rope_sin, const int32_t* rope_pos,
                  int variant, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!q || !k || !v || !d_o || !lse || !delta || !dq || !dk || !dv || bad_geom(B, L, Hq, Hkv, d)) return MM355_EINVAL;
    if ((ld_q & 7) || (ld_k & 7) || (ld_o & 7) || (ld_dkv & 7) || (ld_dq & 7)) return MM355_EINVAL;
    const int64_t ld_max = std::max(std::max(ld_q, ld_k), ld_o);
    // variant: 0 = the product's choice, 2 = attn2.hip (generic d; GQA then needs the 2*B*L*Hq*d workspace), 3 = attn3.hip (two waves per SIMD;
    // -DMM355_LEGACY_VARIANTS builds only), 4 = attn4_bwd.hip (one wave per SIMD, hand-placed streams; needs the workspace), 41 = its
    // serialised debugging streams
    if (variant != 0 && variant != 2 && variant != 3 && variant != 4 && variant != 41) return MM355_EINVAL;
    const bool is128 = fast128(d, ld_max) && variant != 2;
    const bool can4 = is128 && variant != 3 && workspace && L * ld_max * 2 < 0x7fffffffll;
#ifdef MM355_LEGACY_VARIANTS
    const bool fast = is128;                                 // attn3's dK/dV + dQ kernels
#else
    const bool fast = false;
    if (variant == 3) return MM355_EUNSUPPORTED;
    // d == 128 without the stream kernels' workspace / beyond their 31-bit offsets: the generic kernels would need THEIR (larger) GQA workspace,
    // which mm355_attn_bwd_ws_floats did not size for this geometry
    if (is128 && !can4 && Hq != Hkv) return workspace ? MM355_EUNSUPPORTED : MM355_EINVAL;
#endif
    if (rope_cos && !(can4 || fast)) return MM355_EUNSUPPORTED;   // the fused inverse rotation lives in the d == 128 kernels' epilogues
    if ((variant == 4 || variant == 41) && !can4) return MM355_EUNSUPPORTED;
    if (variant == 4 || variant == 41 || (variant == 0 && can4)) {
        attn2::Args a4{q, k, v, d_o, ld_q, ld_k, ld_o, nullptr, nullptr, lse, delta, dq, ld_dq, seqlens,
                       dk, dv, nullptr, nullptr, ld_dkv, (int)B, (int)L, (int)Hq, (int)Hkv, (int)d, scale, causal, rope_cos, rope_sin, rope_pos};
        return mm355_attn4_bwd_launch(a4, workspace, variant == 41, (hipStream_t)stream);
    }
    // GQA on the generic kernels: the group is summed from fp32 partials in the workspace (the d == 128 kernel sums in registers)
    if (Hq != Hkv && !fast && !workspace) return MM355_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* dkp = nullptr;
    float* dvp = nullptr;
    if (Hq != Hkv && !fast) {
        dkp = workspace;
        dvp = workspace + (int64_t)B * L * Hq * d;
    }
    attn2::Args a{q, k, v, d_o, ld_q, ld_k, ld_o, nullptr, nullptr, lse, delta, dq, ld_dq, seqlens,
                  dk, dv, dkp, dvp, ld_dkv, (int)B, (int)L, (int)Hq, (int)Hkv, (int)d, scale, causal, rope_cos, rope_sin, rope_pos};
#ifdef MM355_LEGACY_VARIANTS
    int rc = fast ? mm355_attn3_dkdv_launch(a, s) : mm355_attn2_dkdv_launch(a, pick_dp(d), s);
#else
    int rc = mm355_attn2_dkdv_launch(a, pick_dp(d), s);
#endif
    if (rc != MM355_OK) return rc;
    if (dkp) {
        const int64_t rows = B * L;
        const unsigned rg = (unsigned)std::min<int64_t>((rows * (Hkv * d / 8) + NT - 1) / NT, 4096);
        hipLaunchKernelGGL(group_reduce_kernel, dim3(rg), dim3(NT), 0, s, dkp, dk, ld_dkv, rows, (int)Hq, (int)Hkv, (int)d);
        hipLaunchKernelGGL(group_reduce_kernel, dim3(rg), dim3(NT), 0, s, dvp, dv, ld_dkv, rows, (int)Hq, (int)Hkv, (int)d);
        rc = mm_launch_status();
        if (rc != MM355_OK) return rc;
    }
#ifdef MM355_LEGACY_VARIANTS
    if (fast) return mm355_attn3_dq_launch(a, s);
#endif
    return mm355_attn2_dq_launch(a, pick_dp(d), s);
}
}  // namespace

extern "C" int mm355_attn_bwd(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, const mm355_bf16* d_o,
                              int64_t ld_o, const float* lse, const float* delta, const int32_t* seqlens, mm355_bf16* dq, int64_t ld_dq,
                              mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d,
                              float scale, int causal, float* workspace, void* stream) {
    return attn_bwd_impl(q, k, v, ld_q, ld_k, d_o, ld_o, lse, delta, seqlens, dq, ld_dq, dk, dv, ld_dkv, B, L, Hq, Hkv, d, scale, causal, workspace,
                         nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int mm355_attn_bwd_rope(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, const mm355_bf16* d_o,
                                   int64_t ld_o, const float* lse, const float* delta, const int32_t* seqlens, mm355_bf16* dq, int64_t ld_dq,
                                   mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d,
                                   float scale, int causal, const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* pos_offset,
                                   float* workspace, void* stream) {
    if (!cos_t || !sin_t) return MM355_EINVAL;
    return attn_bwd_impl(q, k, v, ld_q, ld_k, d_o, ld_o, lse, delta, seqlens, dq, ld_dq, dk, dv, ld_dkv, B, L, Hq, Hkv, d, scale, causal, workspace,
                         cos_t, sin_t, pos_offset, 0, stream);
}

extern "C" int mm355_attn_bwd_variant(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, const mm355_bf16* d_o,
                                      int64_t ld_o, const float* lse, const float* delta, const int32_t* seqlens, mm355_bf16* dq, int64_t ld_dq,
                                      mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d,
                                      float scale, int causal, const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* pos_offset,
                                      float* workspace, int variant, void* stream) {
    if ((cos_t == nullptr) != (sin_t == nullptr)) return MM355_EINVAL;
    return attn_bwd_impl(q, k, v, ld_q, ld_k, d_o, ld_o, lse, delta, seqlens, dq, ld_dq, dk, dv, ld_dkv, B, L, Hq, Hkv, d, scale, causal, workspace,
                         cos_t, sin_t, pos_offset, variant, stream);
}
