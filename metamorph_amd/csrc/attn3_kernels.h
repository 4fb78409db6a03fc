// Kernel templates of the d == 128 attention path (instantiated in attn3.hip with RQ = 2: 32 query rows per wave, two waves per SIMD).
// RQ = 4 (64 rows per wave, one wave per SIMD, accumulators in AGPRs: its own translation unit without -amdgpu-mfma-vgpr-form) was built
// and measured in round 2 -- 1.06 vs 0.69 ms forward -- and removed again; numbers in DESIGN.md section 4, profiles/r2_attn_rq_ab.log.
#pragma once
// Attention, d == 128 fast path (gfx950): the formulation of attn2.hip (swapped products on natural [rows][128] tiles, P^T / dS^T
// never leave registers) with the staging rebuilt around the LDS-DMA engine:
//   * K / V tiles of 64 keys go HBM -> LDS with global_load_lds_dwordx4 (1 KiB = 4 tile rows per wave instruction; the XOR
//     swizzle is applied on the per-lane SOURCE address, the LDS side is lane-linear) into a two-deep ring, so a tile costs
//     no VGPRs, no ds_write and ONE barrier; tile t+1 is in flight while tile t is consumed
//   * a wave owns 32 query rows (two 16-column fragments) and re-uses every K / V fragment read for both
//   * the row maximum is all-reduced on the VALU (v_permlane16/32_swap); row sums stay per lane until the epilogue
//   * heavy (late) causal query blocks are launched first
// Replaces torch SDPA as driven by HF LlamaModel (reference call site metamorph_llama.py:349-359).
#include "attn2.h"
#include <type_traits>
#include <cstdlib>

namespace attn3 {
using namespace attn2;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int DP = 128;
constexpr int DS = 128;
constexpr int TILE = 64 * DS * 2;                            // 16 KiB: [64 rows][128] bf16
constexpr float LOG2E = 1.4426950408889634f;
constexpr float M_INIT = -1.0e30f;                           // finite "minus infinity" of the running maximum (log2 domain)

// per-lane source byte offsets of the four 1-KiB pieces a wave moves per tile: piece p = wave*4 + i covers tile rows 4p..4p+3
struct TileSrc {
    uint32_t off[4];
    MM_DEV void init(int wave, int lane, int64_t ld) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 4 + (lane >> 4);
            const int c = (lane & 15) ^ swzN<DS>(row);       // logical chunk that belongs in physical slot (lane & 15)
            off[i] = (uint32_t)(row * ld * 2 + c * 16);
        }
    }
};

// tile rows [row0, row0 + 64) of a [L][ld] matrix -> LDS tile `dst` (this wave's four pieces)
MM_DEV void dma_tile(const uint16_t* base, int64_t ld, int row0, int L, const TileSrc& ts, unsigned char* dst, int wave_s, int lane) {
    unsigned char* d0 = dst + wave_s * 4096;
    if (row0 + 64 <= L) {
        const unsigned char* tb = (const unsigned char*)(base + (int64_t)row0 * ld);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(tb + ts.off[i]), (lptr_t)(d0 + i * 1024), 16, 0, 0);
    } else {                                                 // ragged last tile: clamp the rows (masked by the caller)
        const unsigned char* tb = (const unsigned char*)(base + (int64_t)row0 * ld);
        const int last = L - 1 - row0;
        const uint32_t ldb = (uint32_t)ld * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave_s * 4 + i) * 4 + (lane >> 4);
            const int c = (lane & 15) ^ swzN<DS>(row);
            __builtin_amdgcn_global_load_lds((gptr_t)(tb + (uint32_t)min(row, last) * ldb + (uint32_t)(c * 16)), (lptr_t)(d0 + i * 1024), 16, 0, 0);
        }
    }
}

// 1-D grid -> (x, head, sample) with all blocks that share K / V (fwd, dQ: the query blocks of a GQA group) or Q / dO (dK/dV:
// the key blocks of a query head) on ONE XCD, so the shared tiles stay in that XCD's 4-MiB L2: hardware deals consecutive
// block ids round-robin over the 8 XCDs, so XCD x is given the x-th contiguous eighth of the logical order.
// Logical order: `inner` heads fastest, then x, then the remaining heads, then the sample; `reverse` walks x downwards.  Causal
// work grows with the query block index (fwd, dQ) and shrinks with the key block index (dK/dV): with x SLOWER than a few heads, the
// heavy blocks of several heads start together and the light ones fill the tail.  (x fastest -- 15, 14, ..., 0 per head -- leaves
// the last head's heaviest block to start late: 7 % / 13 % longer kernels in a dispatch simulation with measured block times.)
MM_DEV void block_coords(int nx, int H, int inner, bool reverse, int& x, int& h, int& b) {
    const int total = gridDim.x, bid = blockIdx.x;
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int span = nx * inner, within = logical % span, rest = logical / span, outer = H / inner;
    const int xx = within / inner;
    x = reverse ? nx - 1 - xx : xx;
    h = (rest % outer) * inner + within % inner;
    b = rest / outer;
}
// heads walked fastest: the GQA group (its blocks share K / V), else four heads when they divide evenly
MM_DEV int inner_heads(int H, int group) { return group > 1 ? group : ((H & 3) == 0 ? 4 : ((H & 1) == 0 ? 2 : 1)); }

// One K / V tile of 64 keys for RQ 16-row groups of one wave: S^T = K Q^T, online softmax (log2 domain, deferred rescale), O^T += V^T P^T.
struct FwdLane {
    int fr, fq, k_base, k_x, v_base, v_x;
    MM_DEV void init(int fr_, int fq_) {
        fr = fr_; fq = fq_;
        k_base = fr * (DS * 2); k_x = (fq ^ swzN<DS>(fr)) << 4;
        const int v_row = fq * 4 + (fr >> 2);
        v_base = v_row * (DS * 2) + (fr & 1) * 8; v_x = (((fr & 3) >> 1) ^ swzN<DS>(v_row)) << 4;
    }
};
// ---- the four phases of one K / V tile of 64 keys for RQ 16-row groups of one wave ----
// S^T = K Q^T.  All 16 K fragment reads go out ahead of the MFMAs behind scheduling barriers (left to itself hipcc sinks every ds_read next
// to its use and spends 256 registers on its own prefetch; the pinned form needs 242 and times the same,
// profiles/r2_attn_persist_hoist_ab.log).
template <int RQ>
MM_DEV void fwd_scores(const unsigned char* sK, const FwdLane& ln, const bf16x8 (&qf)[RQ][4], f32x4 (&st)[RQ][4]) {
    constexpr int KS = 4;
    int k_off[4];
    {
        int kx = ln.k_x;
        asm volatile("" : "+v"(kx));                         // keep the derivation inside the tile loop
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) k_off[kk] = (kx ^ (kk << 6)) + ln.k_base;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) st[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 kfr[KS][4];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) kfr[kk][j] = *(const bf16x8*)(sK + j * 4096 + k_off[kk]);
    __builtin_amdgcn_sched_barrier(0);
    // d-step outermost: eight independent accumulator chains, so that back-to-back MFMAs never wait on their own result
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 kf = kfr[kk][j];
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
#ifdef MM355_ABL_NOQK                                        // timing-only ablation builds (tools/): see DESIGN section 4
                asm volatile("" :: "v"(kf));
                st[rq][j][0] += (float)kk;
#else
                st[rq][j] = mfma16(kf, qf[rq][kk], st[rq][j]);
#endif
            }
        }
    __builtin_amdgcn_sched_barrier(0);
}

// mask, row maxima (log2 domain), deferred rescale of the running state
template <int RQ>
MM_DEV void fwd_rowmax(f32x4 (&st)[RQ][4], int kv0, int qw0, int seqlen, int causal, float sl2, const FwdLane& ln,
                       f32x4 (&ot)[RQ][8], float (&m_run)[RQ], f32x4 (&l_part)[RQ]) {
    constexpr int NF = 8;
    constexpr float RESCALE_THR = 6.0f;                      // log2 units: keep the old running max while it grows < 2^6
    const int fr = ln.fr, fq = ln.fq;
    const bool need_mask = (kv0 + 64 > seqlen) || (causal && kv0 + 63 > qw0);
    if (need_mask) {
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
            const int qg = qw0 + rq * 16 + fr;
            const int lim = (causal ? min(qg, seqlen - 1) : seqlen - 1) - kv0 - fq * 4;   // last visible key, tile-relative
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[rq][j][r] = (j * 16 + r > lim) ? -INFINITY : st[rq][j][r];
        }
    }
    float mx[RQ];
    bool grow = false;
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        float m = max3_raw(st[rq][0][0], st[rq][0][1], st[rq][0][2]);                  // 16 values: 8 v_max3 / v_max
        m = max3_raw(m, st[rq][0][3], st[rq][1][0]);
        m = max3_raw(m, st[rq][1][1], st[rq][1][2]);
        m = max3_raw(m, st[rq][1][3], st[rq][2][0]);
        m = max3_raw(m, st[rq][2][1], st[rq][2][2]);
        m = max3_raw(m, st[rq][2][3], st[rq][3][0]);
        m = max3_raw(m, st[rq][3][1], st[rq][3][2]);
        m = max2_raw(m, st[rq][3][3]);
        mx[rq] = quad_max(m) * sl2;                      // max of the RAW scores (scale > 0 commutes with max)
        grow |= mx[rq] > m_run[rq] + RESCALE_THR;
    }
    // deferred rescale: only move the running maxima (and touch the O accumulators) when some row's max grew by more than
    // 2^THR; otherwise P is exponentiated against the old max (bounded by 2^THR).  ONE wave-uniform branch for both row
    // groups, so that everything after it -- V gathers, exponentials, P V -- is a single basic block the scheduler can overlap.
    if (__any(grow)) {
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
            const float mn = fmaxf(m_run[rq], mx[rq]);
            const float alpha = __builtin_amdgcn_exp2f(m_run[rq] - mn);
            l_part[rq] *= alpha;
#pragma unroll
            for (int j = 0; j < NF; ++j) ot[rq][j] *= alpha;
            m_run[rq] = mn;
        }
    }
}

// P = exp2(s * sl2 - m) in place of the scores, row sums; packed fp32 (v_pk_fma_f32 / v_pk_add_f32)
template <int RQ>
MM_DEV void fwd_exp(f32x4 (&st)[RQ][4], float sl2, const float (&m_run)[RQ], f32x4 (&l_part)[RQ]) {
    const f32x4 sl2v = f32x4{sl2, sl2, sl2, sl2};
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const float nm = -m_run[rq];
        const f32x4 nmv = f32x4{nm, nm, nm, nm};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 x = __builtin_elementwise_fma(st[rq][j], sl2v, nmv);
#ifndef MM355_ABL_NOEXP
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
#endif
            st[rq][j] = x;
        }
        l_part[rq] += (st[rq][0] + st[rq][1]) + (st[rq][2] + st[rq][3]);
    }
}

MM_DEV void fwd_v_offsets(const FwdLane& ln, int (&v_off)[8]) {
    int vx = ln.v_x;
    asm volatile("" : "+v"(vx));                             // keep the derivation inside the tile loop
#pragma unroll
    for (int j = 0; j < 8; ++j) v_off[j] = (vx ^ (j << 5)) + ln.v_base;
}

// O^T += V^T P^T; vfr0: the V gathers of the first 32 keys, already issued by the caller
template <int RQ>
MM_DEV void fwd_pv(const unsigned char* sV, const int (&v_off)[8], const bf16x8 (&vfr0)[8], const f32x4 (&st)[RQ][4], f32x4 (&ot)[RQ][8]) {
    constexpr int NF = 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 pb[RQ];
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) pb[rq] = pack_acc(st[rq][2 * kk], st[rq][2 * kk + 1]);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const bf16x8 va = kk == 0 ? vfr0[j] : read_nat_perm_at<DS>(sV, 32, v_off[j]);      // V^T[d][keys perm]
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
#ifdef MM355_ABL_NOPV
                asm volatile("" :: "v"(va), "v"(pb[rq]));
                ot[rq][j][0] += (float)kk;
#else
                ot[rq][j] = mfma16(va, pb[rq], ot[rq][j]);
#endif
            }
        }
    }
}

// one tile, phases in sequence (the one-tile-at-a-time kernels)
template <int RQ>
MM_DEV void fwd_tile(const unsigned char* sK, const unsigned char* sV, int kv0, int qw0, int seqlen, int causal, float sl2, const FwdLane& ln,
                     const bf16x8 (&qf)[RQ][4], f32x4 (&ot)[RQ][8], float (&m_run)[RQ], f32x4 (&l_part)[RQ]) {
    f32x4 st[RQ][4];
    fwd_scores<RQ>(sK, ln, qf, st);
#ifndef MM355_ABL_NOSM
    fwd_rowmax<RQ>(st, kv0, qw0, seqlen, causal, sl2, ln, ot, m_run, l_part);
#endif
    // the V gathers of the first 32 keys go out before the exponentials, which cover their LDS round trip
    int v_off[8];
    fwd_v_offsets(ln, v_off);
    bf16x8 vfr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vfr[j] = read_nat_perm_at<DS>(sV, 0, v_off[j]);
    __builtin_amdgcn_sched_barrier(0);
#ifndef MM355_ABL_NOSM
    fwd_exp<RQ>(st, sl2, m_run, l_part);
#endif
    fwd_pv<RQ>(sV, v_off, vfr, st, ot);
}

// ================================================================================================
// forward: workgroup = 128 query rows (4 waves x 32), KV tiles of 64 keys
// ================================================================================================
template <int RQ>
__global__ __launch_bounds__(256, RQ == 2 ? 2 : 1) void fwd_kernel(Args a) {
    // RQ 16-row groups per wave: RQ = 2 -> 32 query rows per wave, two workgroups per CU (two waves per SIMD) -- the default;
    // RQ = 4 -> 64 rows per wave, ONE wave per SIMD with the whole register file: every K / V fragment read serves four MFMAs
    // instead of two, half the LDS traffic per flop (not the limiter: LDS is 20 % busy, tools/probes/lds_throughput_probe).
    // Measured (B = 12, L = 2048, 32/8 heads): RQ = 4 is SLOWER, 1.06 vs 0.69 ms forward and +0.5 ms in dQ -- with one wave per SIMD
    // nothing overlaps the softmax VALU phase with the MFMA phases, and a compiler-scheduled in-wave pipeline (S(h+1) || softmax(h) ||
    // PV(h-1) in one basic block) was slower still (1.34 ms: ~1 300 accumulator<->VGPR moves).  (Not instantiated any more.)
    constexpr int KS = 4, NF = 8, ROWS = 16 * RQ, BQ = 4 * ROWS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];       // K ring [2] | V ring [2]
#ifdef MM355_ATTN_TIMING                                     // TIMING-ONLY build: phase timestamps of wave 0 overwrite the block's lse rows
    const long long tm0 = __builtin_readcyclecounter();
    long long tm1 = 0, tmd = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int xb, hq, b;
    block_coords((a.L + BQ - 1) / BQ, a.Hq, inner_heads(a.Hq, a.Hq / a.Hkv), true, xb, hq, b);
    const int q0 = xb * BQ;
    const int hk = hq / (a.Hq / a.Hkv);
    const int L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    uint16_t* o_base = a.o + row_base * a.ld_o + (int64_t)hq * DP;
    float* lse_base = a.lse + ((int64_t)b * a.Hq + hq) * L;

    if (q0 >= seqlen) {                                      // whole block is padding: o = 0, lse = 0
        for (int v = tid; v < BQ * (DP / 8); v += 256) {
            const int r = v / (DP / 8), c = (v % (DP / 8)) * 8;
            if (q0 + r < L) *(u32x4*)(o_base + (int64_t)(q0 + r) * a.ld_o + c) = u32x4{0u, 0u, 0u, 0u};
        }
        for (int r = tid; r < BQ; r += 256)
            if (q0 + r < L) lse_base[q0 + r] = 0.f;
        return;
    }

    const int kv_end = a.causal ? min(seqlen, q0 + BQ) : seqlen;
    const int ntiles = (kv_end + 63) >> 6;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * DP;
    const uint16_t* vbase = a.v + row_base * a.ld_k + (int64_t)hk * DP;
    TileSrc ts;
    ts.init(wave, lane, a.ld_k);
    dma_tile(kbase, a.ld_k, 0, L, ts, smem, wave, lane);
    dma_tile(vbase, a.ld_k, 0, L, ts, smem + 2 * TILE, wave, lane);

    const int qw0 = q0 + wave * ROWS;                        // first query row of this wave
    bf16x8 qf[RQ][KS];                                       // B operand: Q[q = fr][d chunk]
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const uint16_t* qp = a.q + (row_base + min(qw0 + rq * 16 + fr, L - 1)) * a.ld_q + (int64_t)hq * DP;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[rq][kk] = *(const bf16x8*)(qp + kk * 32 + fq * 8);
    }
    f32x4 ot[RQ][NF];                                        // O^T[d = j*16 + fq*4 + r][q = fr]
    float m_run[RQ];
    f32x4 l_part[RQ];                                        // this lane's share of the row sum, four partial sums (packed adds)
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        m_run[rq] = M_INIT; l_part[rq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NF; ++j) ot[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // LDS read offsets inside a tile.  K rows (b128): offN(fr, kk*4 + fq) = fr*256 + ((fq ^ swz(fr)) << 4 ^ kk << 6); V gathers (tr_b64):
    // nat_perm_off(j) = row*256 + sub + ((hi ^ swz(row)) << 4 ^ j << 5).  Only the lane-constant halves live across the tile loop (four
    // registers); the per-kk / per-j offsets are one v_xad_u32 each, re-derived per tile behind an opaque copy -- twelve loop-invariant
    // address registers are what hipcc spills first (and a scratch reload carries a vmcnt(0) that serialises the LDS-DMA).
    FwdLane ln;
    ln.init(fr, fq);
    const float sl2 = a.scale * LOG2E;                       // scores in log2 domain: exp2(s*sl2 - m)

    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
#ifndef MM355_ABL_NOSYNC
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my pieces of tile t have landed
        __builtin_amdgcn_s_barrier();                        // everybody's have; everybody is done with tile t-1
        if (t + 1 < ntiles) {
            dma_tile(kbase, a.ld_k, kv0 + 64, L, ts, smem + ((t + 1) & 1) * TILE, wave, lane);
            dma_tile(vbase, a.ld_k, kv0 + 64, L, ts, smem + (2 + ((t + 1) & 1)) * TILE, wave, lane);
        }
#endif
#ifdef MM355_ATTN_TIMING
        if (t == 0) tm1 = __builtin_readcyclecounter();
        if (t == ntiles - 2) tmd = __builtin_readcyclecounter();
#endif
        const unsigned char* sK = smem + (t & 1) * TILE;
        const unsigned char* sV = smem + (2 + (t & 1)) * TILE;
        // a wave whose rows all precede this tile (causal) has nothing to do here
        if (a.causal && kv0 > qw0 + ROWS - 1) continue;
        fwd_tile<RQ>(sK, sV, kv0, qw0, seqlen, a.causal, sl2, ln, qf, ot, m_run, l_part);
    }
#ifdef MM355_ATTN_TIMING
    const long long tm2 = __builtin_readcyclecounter();
#endif
    __syncthreads();                                         // ring is free: reuse it as the output staging area

    // epilogue: O = O^T / l  -> bf16 [q][d] in LDS -> row-contiguous 16-B stores
    unsigned char* so = smem + wave * (ROWS * DP * 2);
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const int qg = qw0 + rq * 16 + fr;
        const bool valid = qg < seqlen;
        const float l_run = quad_sum((l_part[rq][0] + l_part[rq][1]) + (l_part[rq][2] + l_part[rq][3]));
        const float inv = (valid && l_run > 0.f) ? 1.0f / l_run : 0.f;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            u32x2 w;
            w.x = pack2bf(ot[rq][j][0] * inv, ot[rq][j][1] * inv);
            w.y = pack2bf(ot[rq][j][2] * inv, ot[rq][j][3] * inv);
            *(u32x2*)(so + (rq * 16 + fr) * (DP * 2) + (j * 16 + fq * 4) * 2) = w;
        }
        if (fq == 0 && qg < L) lse_base[qg] = valid ? (m_run[rq] + log2f(l_run)) * 0.6931471805599453f : 0.f;
    }
    __syncthreads();
    for (int v = lane; v < ROWS * (DP / 8); v += 64) {
        const int r = v / (DP / 8), c = (v % (DP / 8)) * 8;
        const int qg = qw0 + r;
        if (qg < L) *(u32x4*)(o_base + (int64_t)qg * a.ld_o + c) = *(const u32x4*)(so + r * (DP * 2) + c * 2);
    }
#ifdef MM355_ATTN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tm3 = __builtin_readcyclecounter();
    if (tid == 0) {
        long long* w = (long long*)(lse_base + q0);
        w[0] = tm1 - tm0; w[1] = tmd - tm1; w[2] = tm2 - tmd; w[3] = tm3 - tm2; w[4] = ntiles; w[5] = xb; w[6] = tm0; w[7] = tm3;
    }
#endif
}

// ================================================================================================
// dQ: workgroup = 128 query rows (4 waves x 32); per KV tile S^T = K Q^T and dP^T = V dO^T share the fragment reads of both
// 16-row groups, dS^T feeds dQ^T += K^T dS^T from registers; bf16 result written straight to its column block
// ================================================================================================
template <int RQ>
__global__ __launch_bounds__(256, RQ == 2 ? 2 : 1) void dq_kernel(Args a) {
    constexpr int KS = 4, NF = 8, ROWS = 16 * RQ, BQ = 4 * ROWS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];       // K ring [2] | V ring [2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int xb, hq, b;
    block_coords((a.L + BQ - 1) / BQ, a.Hq, inner_heads(a.Hq, a.Hq / a.Hkv), true, xb, hq, b);
    const int q0 = xb * BQ;
    const int hk = hq / (a.Hq / a.Hkv);
    const int L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    uint16_t* dq_base = a.dqb + row_base * a.ld_dq + (int64_t)hq * DP;

    if (q0 >= seqlen) {                                      // padded query rows carry zero gradient
        for (int v = tid; v < BQ * (DP / 8); v += 256) {
            const int r = v / (DP / 8), c = (v % (DP / 8)) * 8;
            if (q0 + r < L) *(u32x4*)(dq_base + (int64_t)(q0 + r) * a.ld_dq + c) = u32x4{0u, 0u, 0u, 0u};
        }
        return;
    }
    const int kv_end = a.causal ? min(seqlen, q0 + BQ) : seqlen;
    const int ntiles = (kv_end + 63) >> 6;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * DP;
    const uint16_t* vbase = a.v + row_base * a.ld_k + (int64_t)hk * DP;
    TileSrc ts;
    ts.init(wave, lane, a.ld_k);
    dma_tile(kbase, a.ld_k, 0, L, ts, smem, wave, lane);
    dma_tile(vbase, a.ld_k, 0, L, ts, smem + 2 * TILE, wave, lane);

    const int qw0 = q0 + wave * ROWS;
    bf16x8 qf[RQ][KS], dof[RQ][KS];
    float lse2[RQ], del[RQ];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const int qc = min(qw0 + rq * 16 + fr, L - 1);
        const uint16_t* qp = a.q + (row_base + qc) * a.ld_q + (int64_t)hq * DP;
        const uint16_t* dp = a.d_o + (row_base + qc) * a.ld_o + (int64_t)hq * DP;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            qf[rq][kk] = *(const bf16x8*)(qp + kk * 32 + fq * 8);
            dof[rq][kk] = *(const bf16x8*)(dp + kk * 32 + fq * 8);
        }
        lse2[rq] = a.lse_in[((int64_t)b * a.Hq + hq) * L + qc] * LOG2E;            // log2 domain
        del[rq] = a.delta[((int64_t)b * a.Hq + hq) * L + qc];
    }
    f32x4 dqt[RQ][NF];                                       // dQ^T[d = j*16 + fq*4 + r][q = fr]
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
        for (int j = 0; j < NF; ++j) dqt[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int k_off[4];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) k_off[kk] = offN<DS>(fr, kk * 4 + fq);
    int t_off[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) t_off[j] = nat_perm_off<DS>(j, fr, fq);
    const float sl2 = a.scale * LOG2E;
    const f32x4 sl2v = f32x4{sl2, sl2, sl2, sl2}, scv = f32x4{a.scale, a.scale, a.scale, a.scale};
    f32x4 nlse[RQ], ndel[RQ];                                // -lse (log2 domain), -delta * scale
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        nlse[rq] = f32x4{-lse2[rq], -lse2[rq], -lse2[rq], -lse2[rq]};
        const float nd = -del[rq] * a.scale;
        ndel[rq] = f32x4{nd, nd, nd, nd};
    }

    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntiles) {
            dma_tile(kbase, a.ld_k, kv0 + 64, L, ts, smem + ((t + 1) & 1) * TILE, wave, lane);
            dma_tile(vbase, a.ld_k, kv0 + 64, L, ts, smem + (2 + ((t + 1) & 1)) * TILE, wave, lane);
        }
        const unsigned char* sK = smem + (t & 1) * TILE;
        const unsigned char* sV = smem + (2 + (t & 1)) * TILE;
        if (a.causal && kv0 > qw0 + ROWS - 1) continue;

        // per 16-key group j: S^T and dP^T for both row groups, then dS^T = P^T o (dP^T - delta) * scale in place of S^T
        // (dP^T is transient: 8 registers instead of 32)
        const bool need_mask = (kv0 + 64 > seqlen) || (qw0 + ROWS > seqlen) || (a.causal && kv0 + 63 > qw0);
        int lim[RQ];
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
            const int qg = qw0 + rq * 16 + fr;
            const int kmax = qg >= seqlen ? -1 : (a.causal ? min(qg, seqlen - 1) : seqlen - 1);      // last visible key
            lim[rq] = kmax - kv0 - fq * 4;
        }
        f32x4 st[RQ][4];
        auto scores = [&](auto mask_c) {
            constexpr bool MASK = decltype(mask_c)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 dpt[RQ];
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) { st[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dpt[rq] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const bf16x8 kf = *(const bf16x8*)(sK + j * 4096 + k_off[kk]);
                    const bf16x8 vf = *(const bf16x8*)(sV + j * 4096 + k_off[kk]);
#pragma unroll
                    for (int rq = 0; rq < RQ; ++rq) {
                        st[rq][j] = mfma16(kf, qf[rq][kk], st[rq][j]);
                        dpt[rq] = mfma16(vf, dof[rq][kk], dpt[rq]);
                    }
                }
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) {            // dS^T = P^T o (dP^T - delta) * scale, on packed fp32
                    f32x4 p = __builtin_elementwise_fma(st[rq][j], sl2v, nlse[rq]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[r] = __builtin_amdgcn_exp2f(p[r]);
                        if (MASK) p[r] = (j * 16 + r > lim[rq]) ? 0.f : p[r];
                    }
                    st[rq][j] = p * __builtin_elementwise_fma(dpt[rq], scv, ndel[rq]);
                }
            }
        };
        if (need_mask) scores(std::true_type{}); else scores(std::false_type{});
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 sb[RQ];
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) sb[rq] = pack_acc(st[rq][2 * kk], st[rq][2 * kk + 1]);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const bf16x8 ka = read_nat_perm_at<DS>(sK, kk * 32, t_off[j]);        // K^T[d][keys perm]
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) dqt[rq][j] = mfma16(ka, sb[rq], dqt[rq][j]);
            }
        }
    }
    __syncthreads();
    if (a.rope_cos) {
        // inverse RoPE on the bf16-rounded dq (what mm355_rope_qk(inverse) does to the stored gradient): d and d + 64 of a query row are
        // fragments j and j + 4 of the same lane;  dx1 = dy1 c + dy2 s,  dx2 = dy2 c - dy1 s
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
            const int qg = qw0 + rq * 16 + fr;
            const float keep = qg < seqlen ? 1.0f : 0.0f;
            const int pos = min(qg, L - 1) + (a.rope_pos ? a.rope_pos[b] : 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x2 cw = *(const u32x2*)(a.rope_cos + (int64_t)pos * DP + j * 16 + fq * 4);
                const u32x2 sw = *(const u32x2*)(a.rope_sin + (int64_t)pos * DP + j * 16 + fq * 4);
                const float c[4] = {bf2f((uint16_t)(cw.x & 0xffffu)), bf2f((uint16_t)(cw.x >> 16)), bf2f((uint16_t)(cw.y & 0xffffu)), bf2f((uint16_t)(cw.y >> 16))};
                const float sn[4] = {bf2f((uint16_t)(sw.x & 0xffffu)), bf2f((uint16_t)(sw.x >> 16)), bf2f((uint16_t)(sw.y & 0xffffu)), bf2f((uint16_t)(sw.y >> 16))};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x1 = round_bf(dqt[rq][j][r] * keep), x2 = round_bf(dqt[rq][j + 4][r] * keep);
                    dqt[rq][j][r] = x1 * c[r] + x2 * sn[r];
                    dqt[rq][j + 4][r] = x2 * c[r] - x1 * sn[r];
                }
            }
        }
    }
    // dq[q][d] bf16 -> LDS [32 q][128] per wave -> row-contiguous 16-B stores (rows >= seqlen are zero)
    unsigned char* so = smem + wave * (ROWS * DP * 2);
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const float keep = (qw0 + rq * 16 + fr) < seqlen ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            u32x2 w;
            w.x = pack2bf(dqt[rq][j][0] * keep, dqt[rq][j][1] * keep);
            w.y = pack2bf(dqt[rq][j][2] * keep, dqt[rq][j][3] * keep);
            *(u32x2*)(so + (rq * 16 + fr) * (DP * 2) + (j * 16 + fq * 4) * 2) = w;
        }
    }
    __syncthreads();
    for (int v = lane; v < ROWS * (DP / 8); v += 64) {
        const int r = v / (DP / 8), c = (v % (DP / 8)) * 8;
        if (qw0 + r < L) *(u32x4*)(dq_base + (int64_t)(qw0 + r) * a.ld_dq + c) = *(const u32x4*)(so + r * (DP * 2) + c * 2);
    }
}

}  // namespace attn3
