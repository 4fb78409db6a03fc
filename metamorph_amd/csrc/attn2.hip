// Attention, second generation (gfx950): "swapped" formulation on NATURAL tiles.
//
// Every kernel stages only row-major [rows][d] tiles of Q / K / V / dO in LDS (16-B chunk XOR 2*(row&7): conflict-free for
// both access patterns below) and obtains its two kinds of MFMA operands from them:
//   * contraction over d   : ds_read_b128 of a tile row (8 consecutive d)                       -> S^T = K Q^T, dP^T = V dO^T
//   * contraction over rows: ds_read_b64_tr_b16 (hardware 4x16 transpose read) of 4 consecutive rows at one column, in the
//     k-slot permutation of the MFMA accumulator layout, so that P^T / dS^T accumulators are used as the other operand
//     WITHOUT leaving registers                                                                 -> O^T += V^T P^T, dQ^T += K^T dS^T,
//                                                                                                  dV += P^T dO, dK += dS^T Q
// No transposed copies of any tensor exist in HBM, P never goes through LDS, dQ / dK / dV need no atomics:
//   fwd_kernel  : workgroup = 64 query rows (16 per wave), loops over KV tiles of 64 keys, online softmax in log2 domain
//                 with deferred rescale; a query row lives in 4 lanes x 16 registers (2 shuffles per reduction)
//   dq_kernel   : same ownership, three MFMA groups per KV tile
//   dkdv_kernel : workgroup = (KV tile of 64 keys, query head); wave owns 16 keys (K/V fragments in registers), loops over
//                 64-row query tiles; GQA groups are summed afterwards from fp32 partials
// All kernels prefetch the next tile into registers while the current one is consumed.
// Replaces torch SDPA as driven by HF LlamaModel / SiglipAttention (reference call sites metamorph_llama.py:349-359,
// siglip_encoder.py:141).
#include "attn2.h"
#include <cstdlib>

namespace attn2 {

// ---- [64][DP] tile movers with per-thread precomputed byte offsets (tile base is wave-uniform: saddr + voffset loads) ----
template <int DP, int NV>
MM_DEV void init_tile_io(uint32_t (&g)[NV], uint32_t (&sO)[NV], int64_t ld, int d, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT, row = v / (DP / 8), c = v % (DP / 8);
        const bool in = v < Geo<DP>::V64;
        g[i] = (in && c * 8 < d) ? (uint32_t)(row * ld * 2 + c * 16) : 0xffffffffu;
        sO[i] = in ? (uint32_t)offN<Geo<DP>::DS>(row, c) : 0xffffffffu;
    }
}
template <int DP, int NV>
MM_DEV void load_tile(u32x4 (&r)[NV], const uint16_t* base, int64_t ld, int row0, int L, int d, int tid, const uint32_t (&g)[NV]) {
    if (row0 + 64 <= L) {
        const unsigned char* tb = (const unsigned char*)(base + (int64_t)row0 * ld);
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = (g[i] != 0xffffffffu) ? *(const u32x4*)(tb + g[i]) : u32x4{0u, 0u, 0u, 0u};
    } else {                                                 // ragged last tile: clamp the rows (masked by the caller)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * NT, row = v / (DP / 8), c = v % (DP / 8);
            r[i] = u32x4{0u, 0u, 0u, 0u};
            if (v < Geo<DP>::V64 && c * 8 < d) r[i] = *(const u32x4*)(base + (int64_t)min(row0 + row, L - 1) * ld + c * 8);
        }
    }
}
template <int NV>
MM_DEV void store_tile(const u32x4 (&r)[NV], unsigned char* s, const uint32_t (&sO)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (sO[i] != 0xffffffffu) *(u32x4*)(s + sO[i]) = r[i];
}

// ================================================================================================
// forward
// ================================================================================================
template <int DP, int RQ>
__global__ __launch_bounds__(NT) void fwd_kernel(Args a) {
    using G = Geo<DP>;
    constexpr int DS = G::DS, KS = G::KS, NF = G::NF;
    constexpr int NV = (G::V64 + NT - 1) / NT;
    constexpr int ROWS = RQ * 16;                           // query rows per wave
    constexpr int BQ = 4 * ROWS;
    constexpr int EPI = 4 * ROWS * DP * 2;                  // bf16 output staging
    constexpr int SMEM = 2 * G::T64 > EPI ? 2 * G::T64 : EPI;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* sK = smem;
    unsigned char* sV = smem + G::T64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int q0 = blockIdx.x * BQ, hq = blockIdx.y, b = blockIdx.z;
    const int hk = hq / (a.Hq / a.Hkv);
    const int d = a.d, L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    uint16_t* o_base = a.o + row_base * a.ld_o + (int64_t)hq * d;
    float* lse_base = a.lse + ((int64_t)b * a.Hq + hq) * L;

    if (q0 >= seqlen) {                                      // whole tile is padding: o = 0, lse = 0
        for (int v = tid; v < BQ * (d >> 3); v += NT) {
            const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
            if (q0 + r < L) *(u32x4*)(o_base + (int64_t)(q0 + r) * a.ld_o + c) = u32x4{0u, 0u, 0u, 0u};
        }
        for (int r = tid; r < BQ; r += NT)
            if (q0 + r < L) lse_base[q0 + r] = 0.f;
        return;
    }

    const int qw0 = q0 + wave * ROWS;                        // first query row of this wave
    bf16x8 qf[RQ][KS];                                       // B operand: Q[q = fr][d chunk]
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const uint16_t* qp = a.q + (row_base + min(qw0 + rq * 16 + fr, L - 1)) * a.ld_q + (int64_t)hq * d;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = kk * 32 + fq * 8;
            qf[rq][kk] = (c < d) ? *(const bf16x8*)(qp + c) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f32x4 ot[RQ][NF];                                        // O^T[d = j*16 + fq*4 + r][q = fr]
    float m_run[RQ], l_run[RQ];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        m_run[rq] = -INFINITY; l_run[rq] = 0.f;
#pragma unroll
        for (int j = 0; j < NF; ++j) ot[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int kv_end = a.causal ? min(seqlen, q0 + BQ) : seqlen;
    const int ntiles = (kv_end + 63) >> 6;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * d;
    const uint16_t* vbase = a.v + row_base * a.ld_k + (int64_t)hk * d;
    const float sl2 = a.scale * 1.4426950408889634f;         // scores in log2 domain: exp2(s*sl2 - m)

    u32x4 rk[NV], rv[NV];
    uint32_t g[NV], sO[NV];
    init_tile_io<DP, NV>(g, sO, a.ld_k, d, tid);
    load_tile<DP, NV>(rk, kbase, a.ld_k, 0, L, d, tid, g);
    load_tile<DP, NV>(rv, vbase, a.ld_k, 0, L, d, tid, g);
    store_tile<NV>(rk, sK, sO);
    store_tile<NV>(rv, sV, sO);
    __syncthreads();
    constexpr float RESCALE_THR = 6.0f;                      // log2 units: keep the old running max while it grows < 2^6
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        const bool more = t + 1 < ntiles;
        if (more) {
            load_tile<DP, NV>(rk, kbase, a.ld_k, kv0 + 64, L, d, tid, g);
            load_tile<DP, NV>(rv, vbase, a.ld_k, kv0 + 64, L, d, tid, g);
        }
        // a wave whose rows all precede this tile (causal) has nothing to do here
        const bool active = !a.causal || kv0 <= qw0 + ROWS - 1;
        if (active) {
            f32x4 st[RQ][4];
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
                for (int j = 0; j < 4; ++j) st[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const bf16x8 kf = *(const bf16x8*)(sK + offN<DS>(j * 16 + fr, kk * 4 + fq));
#pragma unroll
                    for (int rq = 0; rq < RQ; ++rq) st[rq][j] = mfma16(kf, qf[rq][kk], st[rq][j]);
                }
            const bool need_mask = (kv0 + 64 > seqlen) || (a.causal && kv0 + 63 > qw0);
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
                const int qg = qw0 + rq * 16 + fr;
                if (need_mask) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kg = kv0 + j * 16 + fq * 4 + r;
                            if (!((kg < seqlen) && (!a.causal || kg <= qg))) st[rq][j][r] = -INFINITY;
                        }
                }
                float mx = -INFINITY;                         // max of the RAW scores (scale > 0 commutes with max)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[rq][j][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mx *= sl2;
                // deferred rescale: only move the running max (and touch the O accumulators) when some row's max grew
                // by more than 2^THR; otherwise P is exponentiated against the old max (bounded by 2^THR)
                const bool grow = mx > m_run[rq] + RESCALE_THR || m_run[rq] == -INFINITY;
                if (__any(grow && mx > -INFINITY)) {
                    const float mn = fmaxf(m_run[rq], mx);
                    const float alpha = (m_run[rq] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[rq] - mn);
                    l_run[rq] *= alpha;
#pragma unroll
                    for (int j = 0; j < NF; ++j) ot[rq][j] *= alpha;
                    m_run[rq] = mn;
                }
                const float mref = m_run[rq];
                float rs = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = (mref == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(fmaf(st[rq][j][r], sl2, -mref));
                        st[rq][j][r] = p;
                        rs += p;
                    }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l_run[rq] += rs;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 pb[RQ];
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) pb[rq] = pack_acc(st[rq][2 * kk], st[rq][2 * kk + 1]);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const bf16x8 va = read_nat_perm<DS>(sV, kk * 32, j, fr, fq);      // V^T[d][keys perm]
#pragma unroll
                    for (int rq = 0; rq < RQ; ++rq) ot[rq][j] = mfma16(va, pb[rq], ot[rq][j]);
                }
            }
        }
        __syncthreads();                                     // every wave is done reading this tile
        if (more) {
            store_tile<NV>(rk, sK, sO);
            store_tile<NV>(rv, sV, sO);
            __syncthreads();
        }
    }

    // epilogue: O = O^T / l  -> bf16 [q][d] in LDS -> row-contiguous 16-B stores
    unsigned char* so = smem + wave * (ROWS * DP * 2);
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const int qg = qw0 + rq * 16 + fr;
        const bool valid = qg < seqlen;
        const float inv = (valid && l_run[rq] > 0.f) ? 1.0f / l_run[rq] : 0.f;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            u32x2 w;
            w.x = pack2bf(ot[rq][j][0] * inv, ot[rq][j][1] * inv);
            w.y = pack2bf(ot[rq][j][2] * inv, ot[rq][j][3] * inv);
            *(u32x2*)(so + (rq * 16 + fr) * (DP * 2) + (j * 16 + fq * 4) * 2) = w;
        }
        if (fq == 0 && qg < L) lse_base[qg] = valid ? (m_run[rq] + log2f(l_run[rq])) * 0.6931471805599453f : 0.f;
    }
    __syncthreads();
    for (int v = lane; v < ROWS * (d >> 3); v += 64) {
        const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
        const int qg = qw0 + r;
        if (qg < L) *(u32x4*)(o_base + (int64_t)qg * a.ld_o + c) = *(const u32x4*)(so + r * (DP * 2) + c * 2);
    }
}

// ================================================================================================
// dQ (block owns its query rows; three MFMA groups per KV tile; bf16 result written straight to its column block)
// ================================================================================================
template <int DP>
__global__ __launch_bounds__(NT) void dq_kernel(Args a) {
    using G = Geo<DP>;
    constexpr int DS = G::DS, KS = G::KS, NF = G::NF;
    constexpr int NV = (G::V64 + NT - 1) / NT;
    constexpr int BQ = 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G::T64];
    unsigned char* sK = smem;
    unsigned char* sV = smem + G::T64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int q0 = blockIdx.x * BQ, hq = blockIdx.y, b = blockIdx.z;
    const int hk = hq / (a.Hq / a.Hkv);
    const int d = a.d, L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    uint16_t* dq_base = a.dqb + row_base * a.ld_dq + (int64_t)hq * d;
    const int qw0 = q0 + wave * 16;
    const int qg = qw0 + fr;

    if (q0 >= seqlen) {                                      // padded query rows carry zero gradient
        for (int v = tid; v < BQ * (d >> 3); v += NT) {
            const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
            if (q0 + r < L) *(u32x4*)(dq_base + (int64_t)(q0 + r) * a.ld_dq + c) = u32x4{0u, 0u, 0u, 0u};
        }
        return;
    }
    bf16x8 qf[KS], dof[KS];
    {
        const int qc = min(qg, L - 1);
        const uint16_t* qp = a.q + (row_base + qc) * a.ld_q + (int64_t)hq * d;
        const uint16_t* dp = a.d_o + (row_base + qc) * a.ld_o + (int64_t)hq * d;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = kk * 32 + fq * 8;
            qf[kk] = (c < d) ? *(const bf16x8*)(qp + c) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            dof[kk] = (c < d) ? *(const bf16x8*)(dp + c) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    const float lse_r = a.lse_in[((int64_t)b * a.Hq + hq) * L + min(qg, L - 1)] * 1.4426950408889634f;   // log2 domain
    const float del_r = a.delta[((int64_t)b * a.Hq + hq) * L + min(qg, L - 1)];
    f32x4 dqt[NF];                                           // dQ^T[d = j*16 + fq*4 + r][q = fr]
#pragma unroll
    for (int j = 0; j < NF; ++j) dqt[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kv_end = a.causal ? min(seqlen, q0 + BQ) : seqlen;
    const int ntiles = (kv_end + 63) >> 6;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * d;
    const uint16_t* vbase = a.v + row_base * a.ld_k + (int64_t)hk * d;
    const float sl2 = a.scale * 1.4426950408889634f;

    u32x4 rk[NV], rv[NV];
    uint32_t g[NV], sO[NV];
    init_tile_io<DP, NV>(g, sO, a.ld_k, d, tid);
    load_tile<DP, NV>(rk, kbase, a.ld_k, 0, L, d, tid, g);
    load_tile<DP, NV>(rv, vbase, a.ld_k, 0, L, d, tid, g);
    store_tile<NV>(rk, sK, sO);
    store_tile<NV>(rv, sV, sO);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        const bool more = t + 1 < ntiles;
        if (more) {
            load_tile<DP, NV>(rk, kbase, a.ld_k, kv0 + 64, L, d, tid, g);
            load_tile<DP, NV>(rv, vbase, a.ld_k, kv0 + 64, L, d, tid, g);
        }
        const bool active = !a.causal || kv0 <= qw0 + 15;
        if (active) {
            f32x4 st[4], dpt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { st[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dpt[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const bf16x8 kf = *(const bf16x8*)(sK + offN<DS>(j * 16 + fr, kk * 4 + fq));
                    const bf16x8 vf = *(const bf16x8*)(sV + offN<DS>(j * 16 + fr, kk * 4 + fq));
                    st[j] = mfma16(kf, qf[kk], st[j]);
                    dpt[j] = mfma16(vf, dof[kk], dpt[j]);
                }
            const bool need_mask = (kv0 + 64 > seqlen) || (qw0 + 16 > seqlen) || (a.causal && kv0 + 63 > qw0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(fmaf(st[j][r], sl2, -lse_r));
                    if (need_mask) {
                        const int kg = kv0 + j * 16 + fq * 4 + r;
                        if (!((qg < seqlen) && (kg < seqlen) && (!a.causal || kg <= qg))) p = 0.f;
                    }
                    st[j][r] = p * (dpt[j][r] - del_r) * a.scale;                       // dS^T
                }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 sb = pack_acc(st[2 * kk], st[2 * kk + 1]);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const bf16x8 ka = read_nat_perm<DS>(sK, kk * 32, j, fr, fq);          // K^T[d][keys perm]
                    dqt[j] = mfma16(ka, sb, dqt[j]);
                }
            }
        }
        __syncthreads();                                     // every wave is done reading this tile
        if (more) {
            store_tile<NV>(rk, sK, sO);
            store_tile<NV>(rv, sV, sO);
            __syncthreads();
        }
    }
    // dq[q][d] bf16 -> LDS [16 q][DP] per wave -> row-contiguous 16-B stores (rows >= seqlen are zero)
    unsigned char* so = smem + wave * (16 * DP * 2);
    const float keep = qg < seqlen ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        u32x2 w;
        w.x = pack2bf(dqt[j][0] * keep, dqt[j][1] * keep);
        w.y = pack2bf(dqt[j][2] * keep, dqt[j][3] * keep);
        *(u32x2*)(so + fr * (DP * 2) + (j * 16 + fq * 4) * 2) = w;
    }
    __syncthreads();
    for (int v = lane; v < 16 * (d >> 3); v += 64) {
        const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
        if (qw0 + r < L) *(u32x4*)(dq_base + (int64_t)(qw0 + r) * a.ld_dq + c) = *(const u32x4*)(so + r * (DP * 2) + c * 2);
    }
}

// ================================================================================================
// dK / dV
// ================================================================================================
template <int DP>
__global__ __launch_bounds__(NT) void dkdv_kernel(Args a) {
    using G = Geo<DP>;
    constexpr int DS = G::DS, KS = G::KS, NF = G::NF;
    constexpr int QT = 64;                                   // query rows per iteration
    constexpr int TILE = QT * DS * 2;
    constexpr int NVQ = (QT * (DP / 8) + NT - 1) / NT;
    constexpr int EPI = 4 * 16 * DP * 4;
    constexpr int SMEM = 2 * TILE > EPI ? 2 * TILE : EPI;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    __shared__ __attribute__((aligned(16))) float sStat[2 * QT];           // lse[64] | delta[64] (log2-domain lse)
    unsigned char* sQ = smem;
    unsigned char* sDO = smem + TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int group = a.Hq / a.Hkv;
    const int kv0 = blockIdx.x * 64, hq = blockIdx.y, b = blockIdx.z;
    const int hk = hq / group;
    const int d = a.d, L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    const int mykey0 = kv0 + wave * 16;
    const int64_t ld_p = (int64_t)a.Hq * d;

    auto store_rows = [&](const f32x4 (&acc)[NF], bool is_dv, bool zero) {
        // wave's 16 keys x d -> LDS (fp32) -> row-contiguous stores (fp32 partial or bf16 direct)
        float* so = (float*)smem + wave * (16 * DP);
        if (!zero) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) so[(fq * 4 + r) * DP + j * 16 + fr] = acc[j][r];
        }
        __syncthreads();
        for (int v = lane; v < 16 * (d >> 3); v += 64) {
            const int r = v / (d >> 3), c = (v % (d >> 3)) * 8;
            const int key = mykey0 + r;
            if (key >= L) continue;
            f32x4 x0 = f32x4{0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (!zero) { x0 = *(const f32x4*)(so + r * DP + c); x1 = *(const f32x4*)(so + r * DP + c + 4); }
            if (a.dkp) {
                float* pp = (is_dv ? a.dvp : a.dkp) + (row_base + key) * ld_p + (int64_t)hq * d + c;
                *(f32x4*)pp = x0; *(f32x4*)(pp + 4) = x1;
            } else {
                const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                *(u32x4*)((is_dv ? a.dv : a.dk) + (row_base + key) * a.ld_dkv + (int64_t)hk * d + c) = pack8(f);
            }
        }
        __syncthreads();
    };

    f32x4 dkacc[NF], dvacc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) { dkacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (kv0 >= seqlen) {                                     // keys are all padding: zero gradients
        store_rows(dkacc, false, true);
        store_rows(dvacc, true, true);
        return;
    }

    // K / V fragments of this wave's 16 keys (B operands): lane holds X[key = fr][kk*32 + fq*8 ..]
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
    const int q_start = a.causal ? kv0 : 0;                  // kv0 is a multiple of 64 = QT
    const int n_it = seqlen > q_start ? (seqlen - q_start + QT - 1) / QT : 0;
    const uint16_t* qb = a.q + row_base * a.ld_q + (int64_t)hq * d;
    const uint16_t* dob = a.d_o + row_base * a.ld_o + (int64_t)hq * d;
    const float* lse_b = a.lse_in + ((int64_t)b * a.Hq + hq) * L;
    const float* del_b = a.delta + ((int64_t)b * a.Hq + hq) * L;

    u32x4 pq[NVQ], pdo[NVQ];
    uint32_t gq[NVQ], gdo[NVQ], sq[NVQ];
    float pstat = 0.f;
#pragma unroll
    for (int i = 0; i < NVQ; ++i) {
        const int v = tid + i * NT, r = v / (DP / 8), c = v % (DP / 8);
        const bool ok = v < QT * (DP / 8) && c * 8 < d;
        gq[i] = ok ? (uint32_t)(r * a.ld_q * 2 + c * 16) : 0xffffffffu;
        gdo[i] = ok ? (uint32_t)(r * a.ld_o * 2 + c * 16) : 0xffffffffu;
        sq[i] = (v < QT * (DP / 8)) ? (uint32_t)offN<DS>(r, c) : 0xffffffffu;
    }
    auto fetch = [&](int it) {
        const int qt0 = q_start + it * QT;
        if (qt0 + QT <= L) {
            const unsigned char* qtile = (const unsigned char*)(qb + (int64_t)qt0 * a.ld_q);
            const unsigned char* dtile = (const unsigned char*)(dob + (int64_t)qt0 * a.ld_o);
#pragma unroll
            for (int i = 0; i < NVQ; ++i) {
                pq[i] = (gq[i] != 0xffffffffu) ? *(const u32x4*)(qtile + gq[i]) : u32x4{0u, 0u, 0u, 0u};
                pdo[i] = (gdo[i] != 0xffffffffu) ? *(const u32x4*)(dtile + gdo[i]) : u32x4{0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int i = 0; i < NVQ; ++i) {
                const int v = tid + i * NT, r = v / (DP / 8), c = v % (DP / 8);
                pq[i] = u32x4{0u, 0u, 0u, 0u}; pdo[i] = u32x4{0u, 0u, 0u, 0u};
                if (v < QT * (DP / 8) && c * 8 < d) {
                    const int qrow = min(qt0 + r, L - 1);
                    pq[i] = *(const u32x4*)(qb + (int64_t)qrow * a.ld_q + c * 8);
                    pdo[i] = *(const u32x4*)(dob + (int64_t)qrow * a.ld_o + c * 8);
                }
            }
        }
        if (tid < 2 * QT) {
            const int qrow = min(qt0 + (tid & (QT - 1)), L - 1);
            pstat = (tid < QT) ? lse_b[qrow] * 1.4426950408889634f : del_b[qrow];
        }
    };
    const float sl2 = a.scale * 1.4426950408889634f;
    if (n_it > 0) fetch(0);
    for (int it = 0; it < n_it; ++it) {
        const int qt0 = q_start + it * QT;
        __syncthreads();                                     // previous iteration's LDS reads done
#pragma unroll
        for (int i = 0; i < NVQ; ++i)
            if (sq[i] != 0xffffffffu) { *(u32x4*)(sQ + sq[i]) = pq[i]; *(u32x4*)(sDO + sq[i]) = pdo[i]; }
        if (tid < 2 * QT) sStat[tid] = pstat;
        __syncthreads();
        if (it + 1 < n_it) fetch(it + 1);

        // S[i], dP[i] for the four 16-row fragments: lane holds X[q = i*16 + fq*4 + r][key = fr]
        f32x4 s[4], dp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const bf16x8 qa = *(const bf16x8*)(sQ + offN<DS>(i * 16 + fr, kk * 4 + fq));
                const bf16x8 da = *(const bf16x8*)(sDO + offN<DS>(i * 16 + fr, kk * 4 + fq));
                s[i] = mfma16(qa, kf[kk], s[i]);
                dp[i] = mfma16(da, vf[kk], dp[i]);
            }
        }
        const int kg = mykey0 + fr;
        const bool need_mask = (qt0 + QT > seqlen) || (kv0 + 64 > seqlen) || (a.causal && qt0 < kv0 + 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 l4 = *(const f32x4*)(sStat + i * 16 + fq * 4);
            const f32x4 d4 = *(const f32x4*)(sStat + QT + i * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(fmaf(s[i][r], sl2, -l4[r]));
                if (need_mask) {
                    const int qg = qt0 + i * 16 + fq * 4 + r;
                    if (!((qg < seqlen) && (kg < seqlen) && (!a.causal || kg <= qg))) p = 0.f;
                }
                s[i][r] = p;
                dp[i][r] = p * (dp[i][r] - d4[r]) * a.scale;                  // dS
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 pa = pack_acc(s[2 * ks], s[2 * ks + 1]);
            const bf16x8 dsa = pack_acc(dp[2 * ks], dp[2 * ks + 1]);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const bf16x8 dob8 = read_nat_perm<DS>(sDO, ks * 32, j, fr, fq);
                const bf16x8 qb8 = read_nat_perm<DS>(sQ, ks * 32, j, fr, fq);
                dvacc[j] = mfma16(pa, dob8, dvacc[j]);
                dkacc[j] = mfma16(dsa, qb8, dkacc[j]);
            }
        }
    }
    __syncthreads();
    store_rows(dkacc, false, false);
    store_rows(dvacc, true, false);
}

}  // namespace attn2

template <int RQ>
static int fwd_launch_rq(const attn2::Args& a, int dp, hipStream_t s) {
    using namespace attn2;
    dim3 grid((unsigned)((a.L + 4 * RQ * 16 - 1) / (4 * RQ * 16)), (unsigned)a.Hq, (unsigned)a.B);
    switch (dp) {
        case 64: hipLaunchKernelGGL((fwd_kernel<64, RQ>), grid, dim3(NT), 0, s, a); break;
        case 96: hipLaunchKernelGGL((fwd_kernel<96, RQ>), grid, dim3(NT), 0, s, a); break;
        default: hipLaunchKernelGGL((fwd_kernel<128, RQ>), grid, dim3(NT), 0, s, a); break;
    }
    return mm_launch_status();
}

int mm355_attn2_fwd_launch(const attn2::Args& a, int dp, hipStream_t s) {
    return fwd_launch_rq<1>(a, dp, s);                       // one 16-row fragment per wave (two measured no faster on the tower shapes)
}

int mm355_attn2_dq_launch(const attn2::Args& a, int dp, hipStream_t s) {
    using namespace attn2;
    dim3 grid((unsigned)((a.L + 63) / 64), (unsigned)a.Hq, (unsigned)a.B);
    switch (dp) {
        case 64: hipLaunchKernelGGL((dq_kernel<64>), grid, dim3(NT), 0, s, a); break;
        case 96: hipLaunchKernelGGL((dq_kernel<96>), grid, dim3(NT), 0, s, a); break;
        default: hipLaunchKernelGGL((dq_kernel<128>), grid, dim3(NT), 0, s, a); break;
    }
    return mm_launch_status();
}

int mm355_attn2_dkdv_launch(const attn2::Args& a, int dp, hipStream_t s) {
    using namespace attn2;
    dim3 grid((unsigned)((a.L + 63) / 64), (unsigned)a.Hq, (unsigned)a.B);
    switch (dp) {
        case 64: hipLaunchKernelGGL((dkdv_kernel<64>), grid, dim3(NT), 0, s, a); break;
        case 96: hipLaunchKernelGGL((dkdv_kernel<96>), grid, dim3(NT), 0, s, a); break;
        default: hipLaunchKernelGGL((dkdv_kernel<128>), grid, dim3(NT), 0, s, a); break;
    }
    return mm_launch_status();
}
