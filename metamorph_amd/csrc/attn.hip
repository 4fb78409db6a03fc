// Flash-style attention for gfx950 (fwd + bwd), fp32 online softmax, bf16 MFMA (v_mfma_f32_16x16x32_bf16).
//
// Replaces torch SDPA as driven by HF LlamaModel (causal + key padding, GQA; reference call site
// metamorph_llama.py:349-359) and by HF SiglipAttention (non-causal, d = 72; siglip_encoder.py:141).
//
// Operand plan (every MFMA operand is read as 8 contraction-contiguous bf16 = one ds_read_b128 / b64):
//   fwd   S  = Q K^T    : Q[q][d] (A, registers)      K[key][d]   (B, LDS, row-major as in memory)
//         O += P V      : P[q][key] (A, via LDS)      Vt[d][key]  (B, LDS, from the per-head transpose)
//   bwd   S, dP         : Q/dO[q][d] (A, LDS)         K/V[key][d] (B, registers: the wave's 16 keys)
//         dV += P^T dO  : P^T straight from the S accumulator layout (A)   dOt[d][q] (B, LDS)
//         dK += dS^T Q  : dS^T straight from the accumulator layout (A)    Qt[d][q]  (B, LDS)
//         dQ += dS K    : dS[q][key] (A, via LDS)     Kt[d][key]  (B, LDS) ; fp32 atomics into dq
// The transposed copies (Vt, Qt, Kt, dOt) are produced by mm355_head_transpose / mm355_attn_bwd_prep.
//
// Workgroup = 4 waves.  fwd: 64 query rows (16 per wave) x KV tiles of 64.  bwd: one KV tile of 64 keys
// (16 per wave) of one KV head, looping over the GQA group's query heads and 32-row query tiles.
#include "mm355_common.h"
#include "attn2.h"
#include <cstdlib>
#include <algorithm>

namespace {

constexpr int NT = 256;

MM_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// LDS image of a [rows][DS] bf16 tile: 16-B chunk index XOR (row & MASK); DS = 64 or 128 elements per row.
template <int DS>
MM_DEV int lds_off(int row, int chunk) {
    constexpr int MASK = DS == 64 ? 7 : 15;
    return row * (DS * 2) + ((chunk ^ (row & MASK)) << 4);
}

struct FwdArgs {
    const uint16_t* q; const uint16_t* k; const uint16_t* vt;
    int64_t ld_q, ld_k, ld_o;
    uint16_t* o; float* lse; const int32_t* seqlens;
    int B, L, Lp, Hq, Hkv, d;
    float scale; int causal;
};

// DP = padded head dim (multiple of 32): 64, 96, 128.
template <int DP>
__global__ __launch_bounds__(NT) void attn_fwd_kernel(FwdArgs a) {
    constexpr int DS = DP == 64 ? 64 : 128;                 // LDS row length (elements) of the K tile
    constexpr int KS = DP / 32;                             // k-steps of the QK^T contraction
    constexpr int NF = DP / 16;                             // output fragments along d
    constexpr int K_BYTES = 64 * DS * 2;                    // K tile  [64 keys][DS]
    constexpr int V_BYTES = DP * 128;                       // Vt tile [DP][64 keys]
    constexpr int P_BYTES = 4 * 16 * 128;                   // per wave [16 q][64 keys]
    __shared__ __attribute__((aligned(16))) unsigned char smem[K_BYTES + V_BYTES + P_BYTES];
    unsigned char* sK = smem;
    unsigned char* sV = smem + K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int q0 = blockIdx.x * 64, hq = blockIdx.y, b = blockIdx.z;
    const int hk = hq / (a.Hq / a.Hkv);
    const int d = a.d, L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    unsigned char* sP = smem + K_BYTES + V_BYTES + wave * (16 * 128);

    const int64_t row_base = (int64_t)b * L;
    uint16_t* o_base = a.o + row_base * a.ld_o + (int64_t)hq * d;
    float* lse_base = a.lse + ((int64_t)b * a.Hq + hq) * L;

    if (q0 >= seqlen) {                                      // whole tile is padding: o = 0, lse = 0
        for (int v = tid; v < 64 * (d >> 3); v += NT) {
            const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
            if (q0 + r < L) *(u32x4*)(o_base + (int64_t)(q0 + r) * a.ld_o + c) = u32x4{0u, 0u, 0u, 0u};
        }
        if (tid < 64 && q0 + tid < L) lse_base[q0 + tid] = 0.f;
        return;
    }

    // Q fragments (A operand): lane holds Q[q0 + wave*16 + fr][kk*32 + fq*8 .. +8]
    bf16x8 qf[KS];
    {
        const int qrow = min(q0 + wave * 16 + fr, L - 1);
        const uint16_t* qp = a.q + (row_base + qrow) * a.ld_q + (int64_t)hq * d;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = kk * 32 + fq * 8;
            qf[kk] = (c < d) ? *(const bf16x8*)(qp + c) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    f32x4 oacc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

    const int kv_end = a.causal ? min(seqlen, q0 + 64) : seqlen;
    const int ntiles = (kv_end + 63) >> 6;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * d;
    const uint16_t* vtbase = a.vt + (((int64_t)b * a.Hkv + hk) * d) * a.Lp;

    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        __syncthreads();                                     // previous tile fully consumed
        // K tile: 64 keys x DP (zero beyond d)
        for (int v = tid; v < 64 * (DP / 8); v += NT) {
            const int r = v / (DP / 8), c = v % (DP / 8);
            const int key = min(kv0 + r, L - 1);
            u32x4 val = u32x4{0u, 0u, 0u, 0u};
            if (c * 8 < d) val = *(const u32x4*)(kbase + (int64_t)key * a.ld_k + c * 8);
            *(u32x4*)(sK + lds_off<DS>(r, c)) = val;
        }
        // Vt tile: DP rows (d) x 64 keys (rows >= d zero)
        for (int v = tid; v < DP * 8; v += NT) {
            const int r = v >> 3, c = v & 7;
            u32x4 val = u32x4{0u, 0u, 0u, 0u};
            if (r < d) val = *(const u32x4*)(vtbase + (int64_t)r * a.Lp + kv0 + c * 8);
            *(u32x4*)(sV + lds_off<64>(r, c)) = val;
        }
        __syncthreads();

        // S = Q K^T : 4 fragments of 16 keys
        f32x4 s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(sK + lds_off<DS>(j * 16 + fr, kk * 4 + fq));
                s[j] = mfma16(qf[kk], kf, s[j]);
            }
        }
        // mask + online softmax; lane holds S[q = fq*4 + r][key = j*16 + fr]
        float mloc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qg = q0 + wave * 16 + fq * 4 + r;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kg = kv0 + j * 16 + fr;
                const bool ok = (kg < seqlen) && (!a.causal || kg <= qg);
                const float val = ok ? s[j][r] * a.scale : -INFINITY;
                s[j][r] = val;
                mx = fmaxf(mx, val);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            mloc[r] = mx;
        }
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mn = fmaxf(m_run[r], mloc[r]);
            alpha[r] = (mn == -INFINITY) ? 1.0f : __expf(m_run[r] - mn);
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = (mn == -INFINITY) ? 0.f : __expf(s[j][r] - mn);
                s[j][r] = p;
                rs += p;
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) rs += __shfl_xor(rs, o, 64);
            l_run[r] = l_run[r] * alpha[r] + rs;
            m_run[r] = mn;
        }
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[j][r] *= alpha[r];
        // P -> LDS (bf16) as [16 q][64 keys], then back as the A operand of P V
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = fq * 4 + r, col = j * 16 + fr;
                *(uint16_t*)(sP + lds_off<64>(row, col >> 3) + (col & 7) * 2) = f2bf(s[j][r]);
            }
        __syncthreads();
        bf16x8 pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) pf[kk] = *(const bf16x8*)(sP + lds_off<64>(fr, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 vf = *(const bf16x8*)(sV + lds_off<64>(j * 16 + fr, kk * 4 + fq));
                oacc[j] = mfma16(pf[kk], vf, oacc[j]);
            }
    }

    // epilogue: O / l -> LDS -> row-contiguous stores; lse
    __syncthreads();
    float* so = (float*)smem + wave * (16 * DP);             // [16][DP] fp32 per wave (<= 8 KiB)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qg = q0 + wave * 16 + fq * 4 + r;
        const bool valid = qg < seqlen;
        const float inv = (valid && l_run[r] > 0.f) ? 1.0f / l_run[r] : 0.f;
#pragma unroll
        for (int j = 0; j < NF; ++j) so[(fq * 4 + r) * DP + j * 16 + fr] = oacc[j][r] * inv;
        if (fr == 0 && qg < L) lse_base[qg] = valid ? m_run[r] + __logf(l_run[r]) : 0.f;
    }
    __syncthreads();
    for (int v = lane; v < 16 * (d >> 3); v += 64) {
        const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
        const int qg = q0 + wave * 16 + r;
        if (qg < L) {
            const f32x4 x0 = *(const f32x4*)(so + r * DP + c), x1 = *(const f32x4*)(so + r * DP + c + 4);
            const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            *(u32x4*)(o_base + (int64_t)qg * a.ld_o + c) = pack8(f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward pre-pass: delta[b][h][l] = sum_dd dO*O ; dOt[b][h][dd][l] = dO transposed (row length Lp)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void attn_bwd_prep_kernel(const uint16_t* __restrict__ o, const uint16_t* __restrict__ d_o, int64_t ld_o,
                                                           float* __restrict__ delta, uint16_t* __restrict__ dot, int L, int Lp, int Hq, int d) {
    __shared__ uint16_t tile[64][128 + 2];
    const int l0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int dv = d >> 3;
    // 4 threads per row: delta
    {
        const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
        const int l = l0 + r;
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
    for (int v = threadIdx.x; v < 64 * dv; v += NT) {
        const int r = v / dv, c = (v % dv) * 8;
        const int l = l0 + r;
        uint16_t tmp[8];
        if (l < L) *(u32x4*)tmp = *(const u32x4*)(d_o + ((int64_t)b * L + l) * ld_o + (int64_t)h * d + c);
        else *(u32x4*)tmp = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][c + e] = tmp[e];
    }
    __syncthreads();
    uint16_t* ob = dot + (((int64_t)b * Hq + h) * d) * Lp + l0;
    for (int v = threadIdx.x; v < d * 8; v += NT) {
        const int dd = v >> 3, r = (v & 7) * 8;
        uint16_t tmp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tmp[e] = tile[r + e][dd];
        *(u32x4*)(ob + (int64_t)dd * Lp + r) = *(const u32x4*)tmp;
    }
}

// ------------------------------------------------------------------------------------------------
// backward main kernel
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
    const uint16_t* q; const uint16_t* k; const uint16_t* v; const uint16_t* d_o;
    const uint16_t* qt; const uint16_t* kt; const uint16_t* dot;
    const float* lse; const float* delta; const int32_t* seqlens;
    float* dq; uint16_t* dk; uint16_t* dv;
    float* dkp; float* dvp;          // optional fp32 per-QUERY-head partials [B*L][Hq*d]: grid.y = Hq, summed afterwards
    int64_t ld_q, ld_k, ld_o, ld_dkv;
    int B, L, Lp, Hq, Hkv, d;
    float scale; int causal;
};

template <int DP, bool WITH_DQ>
__global__ __launch_bounds__(NT) void attn_bwd_kernel(BwdArgs a) {
    constexpr int DS = DP == 64 ? 64 : 128;
    constexpr int KS = DP / 32;
    constexpr int NF = DP / 16;
    constexpr int QD_BYTES = 32 * DS * 2;                   // Q / dO tiles [32 q][DS]
    constexpr int T_BYTES = DP * 64;                        // Qt / dOt tiles [DP][32 q]  (64-B rows)
    constexpr int KT_BYTES = DP * 128;                      // Kt tile [DP][64 keys]
    constexpr int DS_BYTES = 32 * 128;                      // dS tile [32 q][64 keys] bf16
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * QD_BYTES + 2 * T_BYTES + KT_BYTES + DS_BYTES];
    __shared__ __attribute__((aligned(16))) float sStat[64];   // lse[32] | delta[32] of the current query tile
    unsigned char* sQ = smem;
    unsigned char* sDO = sQ + QD_BYTES;
    unsigned char* sQt = sDO + QD_BYTES;
    unsigned char* sDOt = sQt + T_BYTES;
    unsigned char* sKt = sDOt + T_BYTES;
    unsigned char* sDS = sKt + KT_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int group = a.Hq / a.Hkv;
    const bool per_qhead = a.dkp != nullptr;
    const int kv0 = blockIdx.x * 64, hk = per_qhead ? (int)blockIdx.y / group : (int)blockIdx.y, b = blockIdx.z;
    const int d = a.d, L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    const int mykey0 = kv0 + wave * 16;                      // this wave's 16 keys

    uint16_t* dk_base = a.dk + row_base * a.ld_dkv + (int64_t)hk * d;
    uint16_t* dv_base = a.dv + row_base * a.ld_dkv + (int64_t)hk * d;

    if (kv0 >= seqlen) {                                     // keys are all padding: zero gradients
        for (int v = tid; v < 64 * (d >> 3); v += NT) {
            const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
            if (kv0 + r < L) {
                if (per_qhead) {
                    const int64_t o = (row_base + kv0 + r) * ((int64_t)a.Hq * d) + (int64_t)blockIdx.y * d + c;
                    *(f32x4*)(a.dkp + o) = f32x4{0.f, 0.f, 0.f, 0.f}; *(f32x4*)(a.dkp + o + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                    *(f32x4*)(a.dvp + o) = f32x4{0.f, 0.f, 0.f, 0.f}; *(f32x4*)(a.dvp + o + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    *(u32x4*)(dk_base + (int64_t)(kv0 + r) * a.ld_dkv + c) = u32x4{0u, 0u, 0u, 0u};
                    *(u32x4*)(dv_base + (int64_t)(kv0 + r) * a.ld_dkv + c) = u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        return;
    }

    // K / V fragments of this wave's 16 keys (B operands of S and dP): lane holds X[key = fr][kk*32 + fq*8..]
    bf16x8 kf[KS], vf[KS];
    {
        const int key = min(mykey0 + fr, L - 1);
        const uint16_t* kp = a.k + (row_base + key) * a.ld_k + (int64_t)hk * d;
        const uint16_t* vp = a.v + (row_base + key) * a.ld_k + (int64_t)hk * d;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = kk * 32 + fq * 8;
            kf[kk] = (c < d) ? *(const bf16x8*)(kp + c) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            vf[kk] = (c < d) ? *(const bf16x8*)(vp + c) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // Kt tile [DP][64 keys] (B operand of dQ = dS K), loaded once
    if constexpr (WITH_DQ) {
        const uint16_t* ktb = a.kt + (((int64_t)b * a.Hkv + hk) * d) * a.Lp + kv0;
        for (int v = tid; v < DP * 8; v += NT) {
            const int r = v >> 3, c = v & 7;
            u32x4 val = u32x4{0u, 0u, 0u, 0u};
            if (r < d) val = *(const u32x4*)(ktb + (int64_t)r * a.Lp + c * 8);
            *(u32x4*)(sKt + lds_off<64>(r, c)) = val;
        }
    }

    f32x4 dkacc[NF], dvacc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) { dkacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int q_start = a.causal ? (kv0 & ~31) : 0;          // first 32-row query tile that can see this KV tile
    const int q_stop = seqlen;                               // padded query rows carry zero gradient

    const int g_lo = per_qhead ? (int)blockIdx.y % group : 0, g_hi = per_qhead ? g_lo + 1 : group;
    // flattened (query head, 32-row query tile) iteration space with register-staged prefetch: the global loads of
    // iteration it+1 are in flight while iteration it computes (T14: issue early, write LDS late)
    const int ntq = q_stop > q_start ? (q_stop - q_start + 31) / 32 : 0;
    const int n_it = (g_hi - g_lo) * ntq;
    constexpr int NVQ = (32 * (DP / 8) + NT - 1) / NT, NVT2 = (DP * 4 + NT - 1) / NT;
    u32x4 pq[NVQ], pdo[NVQ], pqt[NVT2], pdot[NVT2];
    float pstat = 0.f;
    auto fetch = [&](int it) {
        const int hq = hk * group + g_lo + it / ntq;
        const int qt0 = q_start + (it % ntq) * 32;
        const uint16_t* qb = a.q + row_base * a.ld_q + (int64_t)hq * d;
        const uint16_t* dob = a.d_o + row_base * a.ld_o + (int64_t)hq * d;
        const uint16_t* qtb = a.qt + (((int64_t)b * a.Hq + hq) * d) * a.Lp;
        const uint16_t* dotb = a.dot + (((int64_t)b * a.Hq + hq) * d) * a.Lp;
#pragma unroll
        for (int i = 0; i < NVQ; ++i) {
            const int v = tid + i * NT, r = v / (DP / 8), c = v % (DP / 8);
            pq[i] = u32x4{0u, 0u, 0u, 0u}; pdo[i] = u32x4{0u, 0u, 0u, 0u};
            if (v < 32 * (DP / 8) && c * 8 < d) {
                const int qrow = min(qt0 + r, L - 1);
                pq[i] = *(const u32x4*)(qb + (int64_t)qrow * a.ld_q + c * 8);
                pdo[i] = *(const u32x4*)(dob + (int64_t)qrow * a.ld_o + c * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < NVT2; ++i) {
            const int v = tid + i * NT, r = v >> 2, c = v & 3;
            pqt[i] = u32x4{0u, 0u, 0u, 0u}; pdot[i] = u32x4{0u, 0u, 0u, 0u};
            if (v < DP * 4 && r < d) {
                pqt[i] = *(const u32x4*)(qtb + (int64_t)r * a.Lp + qt0 + c * 8);
                pdot[i] = *(const u32x4*)(dotb + (int64_t)r * a.Lp + qt0 + c * 8);
            }
        }
        if (tid < 64) {
            const int qrow = min(qt0 + (tid & 31), L - 1);
            pstat = (tid < 32) ? a.lse[((int64_t)b * a.Hq + hq) * L + qrow] : a.delta[((int64_t)b * a.Hq + hq) * L + qrow];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NVQ; ++i) {
            const int v = tid + i * NT, r = v / (DP / 8), c = v % (DP / 8);
            if (v < 32 * (DP / 8)) {
                *(u32x4*)(sQ + lds_off<DS>(r, c)) = pq[i];
                *(u32x4*)(sDO + lds_off<DS>(r, c)) = pdo[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NVT2; ++i) {
            const int v = tid + i * NT, r = v >> 2, c = v & 3;
            if (v < DP * 4) {
                const int off = r * 64 + ((c ^ ((r >> 2) & 3)) << 4);   // rows r, r+4, r+8, r+12 share a bank window
                *(u32x4*)(sQt + off) = pqt[i];
                *(u32x4*)(sDOt + off) = pdot[i];
            }
        }
        if (tid < 64) sStat[tid] = pstat;
    };
    if (n_it > 0) fetch(0);
    for (int it = 0; it < n_it; ++it) {
        {
            const int hq = hk * group + g_lo + it / ntq;
            const int qt0 = q_start + (it % ntq) * 32;
            float* dq_b = a.dq + row_base * ((int64_t)a.Hq * d) + (int64_t)hq * d;
            (void)dq_b;
            __syncthreads();                                 // previous iteration's LDS reads done
            commit();
            __syncthreads();
            if (it + 1 < n_it) fetch(it + 1);

            // S[i] / dP[i] for the two 16-row halves i: lane holds X[q = i*16 + fq*4 + r][key = fr]
            f32x4 s[2], dp[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const bf16x8 qa = *(const bf16x8*)(sQ + lds_off<DS>(i * 16 + fr, kk * 4 + fq));
                    const bf16x8 da = *(const bf16x8*)(sDO + lds_off<DS>(i * 16 + fr, kk * 4 + fq));
                    s[i] = mfma16(qa, kf[kk], s[i]);
                    dp[i] = mfma16(da, vf[kk], dp[i]);
                }
            }
            // P = exp(S*scale - lse), dS = P * (dP - delta) * scale ; masked entries -> 0
            const int kg = mykey0 + fr;
            f32x4 lse4[2], del4[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                lse4[i] = *(const f32x4*)(sStat + i * 16 + fq * 4);
                del4[i] = *(const f32x4*)(sStat + 32 + i * 16 + fq * 4);
            }
            uint16_t pbits[2][4], dsbits[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qg = qt0 + i * 16 + fq * 4 + r;
                    const bool ok = (qg < seqlen) && (kg < seqlen) && (!a.causal || kg <= qg);
                    float p = 0.f, dsv = 0.f;
                    if (ok) {
                        p = __expf(s[i][r] * a.scale - lse4[i][r]);
                        dsv = p * (dp[i][r] - del4[i][r]) * a.scale;
                    }
                    pbits[i][r] = f2bf(p);
                    dsbits[i][r] = f2bf(dsv);
                }
            // A operands straight from the accumulator layout: lane (key = fr, fq) holds q = fq*4 + r (i = 0)
            // and 16 + fq*4 + r (i = 1): 8 contraction values.  The matching B operand reads the same 8 q's.
            bf16x8 pa, dsa;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pa[i * 4 + r] = (short)pbits[i][r]; dsa[i * 4 + r] = (short)dsbits[i][r]; }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                // B[k = q][n = dcol]: lane (dcol = j*16 + fr, fq) reads Xt[dcol][fq*4 .. +4] and Xt[dcol][16 + fq*4 .. +4]
                const int row = j * 16 + fr;
                const int c0 = (fq >> 1), c1 = 2 + (fq >> 1);   // 16-B chunk holding q = fq*4 (resp. 16 + fq*4)
                const int sub = (fq & 1) * 8;                    // byte offset of the 4-element half inside the chunk
                const int o0 = row * 64 + ((c0 ^ ((row >> 2) & 3)) << 4) + sub;
                const int o1 = row * 64 + ((c1 ^ ((row >> 2) & 3)) << 4) + sub;
                const bf16x4 d0 = *(const bf16x4*)(sDOt + o0), d1 = *(const bf16x4*)(sDOt + o1);
                const bf16x4 q0v = *(const bf16x4*)(sQt + o0), q1v = *(const bf16x4*)(sQt + o1);
                const bf16x8 dob8 = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
                const bf16x8 qb8 = {q0v[0], q0v[1], q0v[2], q0v[3], q1v[0], q1v[1], q1v[2], q1v[3]};
                dvacc[j] = mfma16(pa, dob8, dvacc[j]);
                dkacc[j] = mfma16(dsa, qb8, dkacc[j]);
            }
            if constexpr (WITH_DQ) {
            // dS -> LDS as [32 q][64 keys] for dQ = dS K (contraction over the workgroup's 64 keys)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i * 16 + fq * 4 + r, col = wave * 16 + fr;
                    *(uint16_t*)(sDS + lds_off<64>(row, col >> 3) + (col & 7) * 2) = dsbits[i][r];
                }
            __syncthreads();
            // each wave: dQ[32 q][columns wave*DP/4 .. +DP/4]
            constexpr int NQ = DP / 64;                       // 16-col fragments per wave (1 for 64, 2 for 128)
            constexpr int NQF = NQ > 0 ? NQ : 1;
            if constexpr (DP == 96) {
                // 6 column fragments spread over 4 waves: waves 0,1 take two, waves 2,3 take one
            }
            const int nfrag = (DP == 96) ? (wave < 2 ? 2 : 1) : NQF;
            const int frag0 = (DP == 96) ? (wave < 2 ? wave * 2 : 2 + wave) : wave * NQF;
            for (int jf = 0; jf < nfrag; ++jf) {
                const int j = frag0 + jf;                     // column fragment: dcols j*16 .. +16
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const bf16x8 av = *(const bf16x8*)(sDS + lds_off<64>(i * 16 + fr, kk * 4 + fq));
                        const bf16x8 bv = *(const bf16x8*)(sKt + lds_off<64>(j * 16 + fr, kk * 4 + fq));
                        acc = mfma16(av, bv, acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qg = qt0 + i * 16 + fq * 4 + r, col = j * 16 + fr;
                        if (qg < seqlen && col < d) atomicAdd(dq_b + (int64_t)qg * ((int64_t)a.Hq * d) + col, acc[r]);
                    }
                }
            }
            }  // WITH_DQ
        }
    }

    // epilogue: dK, dV of this wave's 16 keys -> LDS -> row stores
    __syncthreads();
    float* so = (float*)smem + wave * (16 * DP);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(fq * 4 + r) * DP + j * 16 + fr] = pass ? dvacc[j][r] : dkacc[j][r];
        __syncthreads();
        uint16_t* ob = pass ? dv_base : dk_base;
        for (int v = lane; v < 16 * (d >> 3); v += 64) {
            const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
            const int key = mykey0 + r;
            if (key < L) {
                const f32x4 x0 = *(const f32x4*)(so + r * DP + c), x1 = *(const f32x4*)(so + r * DP + c + 4);
                if (per_qhead) {
                    float* pp = (pass ? a.dvp : a.dkp) + (row_base + key) * ((int64_t)a.Hq * d) + (int64_t)blockIdx.y * d + c;
                    *(f32x4*)pp = x0; *(f32x4*)(pp + 4) = x1;
                } else {
                    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    *(u32x4*)(ob + (int64_t)key * a.ld_dkv + c) = pack8(f);
                }
            }
        }
    }
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

// MM355_ATTN_V1=1 selects the first-generation kernels of this file (kept for A/B runs)
bool use_v1() {
    const char* e = std::getenv("MM355_ATTN_V1");
    return e && e[0] == '1';
}

}  // namespace

extern "C" int mm355_attn_fwd(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* vt, int64_t ld_q, int64_t ld_k, mm355_bf16* o,
                              int64_t ld_o, float* lse, const int32_t* seqlens, int64_t B, int64_t L, int64_t Lp, int64_t Hq, int64_t Hkv,
                              int64_t d, float scale, int causal, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!q || !k || !vt || !o || !lse || B <= 0 || L <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || d <= 0 || d > 128 || (d & 7)) return MM355_EINVAL;
    if ((ld_q & 7) || (ld_k & 7) || (ld_o & 7) || (Lp & 63) || Lp < L) return MM355_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (!use_v1()) {
        attn2::Args a2{q, k, nullptr, vt, nullptr, nullptr, ld_q, ld_k, ld_o, o, lse, nullptr, nullptr, nullptr, seqlens,
                       (int)B, (int)L, (int)Lp, (int)Hq, (int)Hkv, (int)d, scale, causal};
        return mm355_attn2_fwd_launch(a2, pick_dp(d), s);
    }
    FwdArgs a{q, k, vt, ld_q, ld_k, ld_o, o, lse, seqlens, (int)B, (int)L, (int)Lp, (int)Hq, (int)Hkv, (int)d, scale, causal};
    dim3 grid((unsigned)((L + 63) / 64), (unsigned)Hq, (unsigned)B);
    switch (pick_dp(d)) {
        case 64: hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(NT), 0, s, a); break;
        case 96: hipLaunchKernelGGL(attn_fwd_kernel<96>, grid, dim3(NT), 0, s, a); break;
        default: hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(NT), 0, s, a); break;
    }
    return mm_launch_status();
}

extern "C" int mm355_attn_bwd_prep(const mm355_bf16* o, const mm355_bf16* d_o, int64_t ld_o, float* delta, mm355_bf16* dot, int64_t B, int64_t L,
                                   int64_t Lp, int64_t Hq, int64_t d, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!o || !d_o || !delta || !dot || B <= 0 || L <= 0 || Hq <= 0 || d <= 0 || d > 128 || (d & 7) || (ld_o & 7) || (Lp & 63) || Lp < L) return MM355_EINVAL;
    dim3 grid((unsigned)(Lp / 64), (unsigned)Hq, (unsigned)B);
    hipLaunchKernelGGL(attn_bwd_prep_kernel, grid, dim3(NT), 0, (hipStream_t)stream, o, d_o, ld_o, delta, dot, (int)L, (int)Lp, (int)Hq, (int)d);
    return mm_launch_status();
}

extern "C" int mm355_attn_bwd(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k, const mm355_bf16* d_o,
                              int64_t ld_o, const mm355_bf16* qt, const mm355_bf16* kt, const mm355_bf16* dot, const float* lse,
                              const float* delta, const int32_t* seqlens, float* dq_f32, mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv,
                              int64_t B, int64_t L, int64_t Lp, int64_t Hq, int64_t Hkv, int64_t d, float scale, int causal,
                              float* workspace, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!q || !k || !v || !d_o || !qt || !kt || !dot || !lse || !delta || !dq_f32 || !dk || !dv) return MM355_EINVAL;
    if (B <= 0 || L <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || d <= 0 || d > 128 || (d & 7)) return MM355_EINVAL;
    if ((ld_q & 7) || (ld_k & 7) || (ld_o & 7) || (ld_dkv & 7) || (Lp & 63) || Lp < L) return MM355_EINVAL;
    BwdArgs a{q, k, v, d_o, qt, kt, dot, lse, delta, seqlens, dq_f32, dk, dv, nullptr, nullptr, ld_q, ld_k, ld_o, ld_dkv,
              (int)B, (int)L, (int)Lp, (int)Hq, (int)Hkv, (int)d, scale, causal};
    dim3 grid((unsigned)((L + 63) / 64), (unsigned)Hkv, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (!use_v1()) {
        // dK/dV: KV-tile-owning kernel without the dQ part; dQ: query-tile-owning kernel (no atomics).
        // With GQA and a workspace the dK/dV kernel runs one workgroup per (KV tile, QUERY head) -- group x more,
        // better balanced workgroups -- writing fp32 partials that a small kernel sums over the group.
        dim3 g2 = grid;
        if (workspace && Hq != Hkv) {
            a.dkp = workspace;
            a.dvp = workspace + (int64_t)B * L * Hq * d;
            g2 = dim3((unsigned)((L + 63) / 64), (unsigned)Hq, (unsigned)B);
        }
        switch (pick_dp(d)) {
            case 64: hipLaunchKernelGGL((attn_bwd_kernel<64, false>), g2, dim3(NT), 0, s, a); break;
            case 96: hipLaunchKernelGGL((attn_bwd_kernel<96, false>), g2, dim3(NT), 0, s, a); break;
            default: hipLaunchKernelGGL((attn_bwd_kernel<128, false>), g2, dim3(NT), 0, s, a); break;
        }
        int rc = mm_launch_status();
        if (rc != MM355_OK) return rc;
        if (a.dkp) {
            const int64_t rows = B * L;
            const unsigned rg = (unsigned)std::min<int64_t>((rows * (Hkv * d / 8) + NT - 1) / NT, 4096);
            hipLaunchKernelGGL(group_reduce_kernel, dim3(rg), dim3(NT), 0, s, a.dkp, dk, ld_dkv, rows, (int)Hq, (int)Hkv, (int)d);
            hipLaunchKernelGGL(group_reduce_kernel, dim3(rg), dim3(NT), 0, s, a.dvp, dv, ld_dkv, rows, (int)Hq, (int)Hkv, (int)d);
            rc = mm_launch_status();
            if (rc != MM355_OK) return rc;
        }
        attn2::Args a2{q, k, v, nullptr, kt, d_o, ld_q, ld_k, ld_o, nullptr, nullptr, lse, delta, dq_f32, seqlens,
                       (int)B, (int)L, (int)Lp, (int)Hq, (int)Hkv, (int)d, scale, causal};
        return mm355_attn2_dq_launch(a2, pick_dp(d), s);
    }
    switch (pick_dp(d)) {
        case 64: hipLaunchKernelGGL((attn_bwd_kernel<64, true>), grid, dim3(NT), 0, s, a); break;
        case 96: hipLaunchKernelGGL((attn_bwd_kernel<96, true>), grid, dim3(NT), 0, s, a); break;
        default: hipLaunchKernelGGL((attn_bwd_kernel<128, true>), grid, dim3(NT), 0, s, a); break;
    }
    return mm_launch_status();
}
