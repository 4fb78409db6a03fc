// Attention, d == 128 fast path (gfx950): launchers + the non-template kernels (ping-pong forward experiment, dK/dV).
// The forward / dQ kernel templates live in attn3_kernels.h.
#include "attn3_kernels.h"

namespace attn3 {
using namespace attn2;

// ================================================================================================
// forward, ping-pong form: workgroup = 256 query rows (8 waves x 32) = two groups (wave w and w + 4 share a SIMD) running ONE
// BARRIER APART.  Per KV tile a wave alternates an M phase -- the 32 PV MFMAs of tile t-1 and the 32 QK^T MFMAs of tile t,
// with their fragment reads -- and a V phase -- the softmax of tile t (all VALU: exponentials, running max / sum, packing
// P^T), its four LDS-DMA pieces of tile t+2 and the counted wait for tile t+1; while one group is in M the other is in V, so
// the matrix pipe and the VALU of every SIMD are fed by different waves instead of fighting inside one.
// K / V tiles live in four-slot rings (128 KiB, 1 workgroup per CU); K(t+3) and V(t+2) are issued in V(t), the counted wait
// for K(t+1) / V(t) sits at the end of M(t) so that BOTH groups have retired their pieces one barrier before anyone reads them.
// EXPERIMENT (MM355_ATTN_PP=1), parity-green but 10 % SLOWER than fwd_kernel at B=8, L=2048 (0.52 vs 0.47 ms): unlike the GEMM,
// the M phase still carries its fragment reads and 256-row causal blocks idle more waves per tile; kept for A/B only.
// ================================================================================================
__global__ __launch_bounds__(512) void fwd_pp_kernel(Args a) {
    constexpr int KS = 4, NF = 8, RQ = 2, ROWS = 32, BQ = 256, NSLOT = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // K ring [4] | V ring [4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
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
        for (int v = tid; v < BQ * (DP / 8); v += 512) {
            const int r = v / (DP / 8), c = (v % (DP / 8)) * 8;
            if (q0 + r < L) *(u32x4*)(o_base + (int64_t)(q0 + r) * a.ld_o + c) = u32x4{0u, 0u, 0u, 0u};
        }
        for (int r = tid; r < BQ; r += 512)
            if (q0 + r < L) lse_base[q0 + r] = 0.f;
        return;
    }
    const int kv_end = a.causal ? min(seqlen, q0 + BQ) : seqlen;
    const int ntiles = (kv_end + 63) >> 6;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * DP;
    const uint16_t* vbase = a.v + row_base * a.ld_k + (int64_t)hk * DP;
    // LDS-DMA duty: 16 K pieces + 16 V pieces per tile over 8 waves: wave w moves K pieces 2w, 2w+1 and the same of V
    uint32_t soff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 4 + (lane >> 4);
        soff[i] = (uint32_t)(row * a.ld_k * 2 + (((lane & 15) ^ swzN<DS>(row)) << 4));
    }
    auto issue = [&](const uint16_t* base, int t, int ring) {   // rows of tile t of K (ring 0) or V (ring 1) -> slot t & 3
        const int row0 = t * 64;
        unsigned char* dst = smem + (ring * NSLOT + (t & 3)) * TILE + wave * 2048;
        if (row0 + 64 <= L) {
            const unsigned char* tb = (const unsigned char*)(base + (int64_t)row0 * a.ld_k);
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(tb + soff[i]), (lptr_t)(dst + i * 1024), 16, 0, 0);
        } else {                                             // ragged last tile: clamp the rows (masked below)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave * 2 + i) * 4 + (lane >> 4);
                const int64_t off = (int64_t)min(row0 + row, L - 1) * a.ld_k + (((lane & 15) ^ swzN<DS>(row)) << 3);
                __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(dst + i * 1024), 16, 0, 0);
            }
        }
    };
    auto wait_pieces = [&](int n) {                          // at most n of my LDS-DMA pieces may still be in flight
        if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // K runs three tiles ahead of its use, V two (V(t) is consumed one phase later than K(t)); issue order = need order
    issue(kbase, 0, 0);
    if (1 < ntiles) issue(kbase, 1, 0);
    issue(vbase, 0, 1);
    if (2 < ntiles) issue(kbase, 2, 0);
    if (1 < ntiles) issue(vbase, 1, 1);

    const int qw0 = q0 + wave * ROWS;                        // first query row of this wave
    bf16x8 qf[RQ][KS];                                       // B operand: Q[q = fr][d chunk]
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const uint16_t* qp = a.q + (row_base + min(qw0 + rq * 16 + fr, L - 1)) * a.ld_q + (int64_t)hq * DP;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[rq][kk] = *(const bf16x8*)(qp + kk * 32 + fq * 8);
    }
    f32x4 ot[RQ][NF];                                        // O^T[d = j*16 + fq*4 + r][q = fr]
    float m_run[RQ], l_part[RQ];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        m_run[rq] = M_INIT; l_part[rq] = 0.f;
#pragma unroll
        for (int j = 0; j < NF; ++j) ot[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int k_off[4];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) k_off[kk] = offN<DS>(fr, kk * 4 + fq);
    const float sl2 = a.scale * LOG2E;
    constexpr float RESCALE_THR = 6.0f;
    // a wave has work on tile t iff some of its rows may see it
    auto active = [&](int t) { return t >= 0 && t < ntiles && !(a.causal && t * 64 > qw0 + ROWS - 1); };

    f32x4 st[RQ][4];                                         // S^T of the tile between its M and V phase
    bf16x8 pb[RQ][2];                                        // P^T (packed) between V(t) and M(t+1)

    wait_pieces(2 * ((1 < ntiles) + 1 + (2 < ntiles) + (1 < ntiles)));   // everything but K(0) may still fly (Q loads drain too)
    __builtin_amdgcn_s_barrier();                            // K(0) is in LDS for everybody
    if (grp == 1) __builtin_amdgcn_s_barrier();              // the second group runs one barrier behind

    for (int t = 0; t <= ntiles; ++t) {
        // ---------------- M phase: PV of tile t-1, then QK^T of tile t
        __builtin_amdgcn_s_setprio(1);
        if (active(t - 1)) {
            const unsigned char* sV = smem + (NSLOT + ((t - 1) & 3)) * TILE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const bf16x8 va = read_nat_perm<DS>(sV, kk * 32, j, fr, fq);      // V^T[d][keys perm]
#pragma unroll
                    for (int rq = 0; rq < RQ; ++rq) ot[rq][j] = mfma16(va, pb[rq][kk], ot[rq][j]);
                }
        }
        if (active(t)) {
            const unsigned char* sK = smem + (t & 3) * TILE;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) st[rq][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8 kf = *(const bf16x8*)(sK + j * 4096 + k_off[kk]);
#pragma unroll
                    for (int rq = 0; rq < RQ; ++rq) st[rq][j] = mfma16(kf, qf[rq][kk], st[rq][j]);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        // my reads of this phase are done before anybody restages those slots; K(t+1) and V(t), which the next M phase of
        // EITHER group reads, have landed (only the batch issued in V(t-1) -- K(t+2), V(t+1) -- may still fly)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_pieces(2 * ((t + 2 < ntiles) + (t + 1 < ntiles)));
        __builtin_amdgcn_s_barrier();
        // ---------------- V phase: stage K(t+3), V(t+2); softmax of tile t
        if (t + 3 < ntiles) issue(kbase, t + 3, 0);
        if (t + 2 < ntiles) issue(vbase, t + 2, 1);
        if (active(t)) {
            const int kv0 = t * 64;
            const bool need_mask = (kv0 + 64 > seqlen) || (a.causal && kv0 + 63 > qw0);
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
                if (need_mask) {
                    const int qg = qw0 + rq * 16 + fr;
                    const int lim = (a.causal ? min(qg, seqlen - 1) : seqlen - 1) - kv0 - fq * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[rq][j][r] = (j * 16 + r > lim) ? -INFINITY : st[rq][j][r];
                }
                float mx = fmaxf(fmaxf(st[rq][0][0], st[rq][0][1]), fmaxf(st[rq][0][2], st[rq][0][3]));
#pragma unroll
                for (int j = 1; j < 4; ++j) mx = fmaxf(mx, fmaxf(fmaxf(st[rq][j][0], st[rq][j][1]), fmaxf(st[rq][j][2], st[rq][j][3])));
                mx = quad_max(mx) * sl2;
                if (__any(mx > m_run[rq] + RESCALE_THR)) {   // deferred rescale (covers the PV of tile t-1 just accumulated)
                    const float mn = fmaxf(m_run[rq], mx);
                    const float alpha = __builtin_amdgcn_exp2f(m_run[rq] - mn);
                    l_part[rq] *= alpha;
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
                        const float p = __builtin_amdgcn_exp2f(fmaf(st[rq][j][r], sl2, -mref));
                        st[rq][j][r] = p;
                        rs += p;
                    }
                l_part[rq] += rs;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) pb[rq][kk] = pack_acc(st[rq][2 * kk], st[rq][2 * kk + 1]);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();              // the first group catches the barrier count up
    __syncthreads();                                         // ring is free: reuse it as the output staging area

    unsigned char* so = smem + wave * (ROWS * DP * 2);
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        const int qg = qw0 + rq * 16 + fr;
        const bool valid = qg < seqlen;
        const float l_run = quad_sum(l_part[rq]);
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
}

// ================================================================================================
// dK / dV: workgroup = (KV tile of 64 keys, KV head); wave owns 16 keys (K / V fragments in registers) and walks the 64-row
// query tiles of every query head of its GQA group, whose Q / dO rows and lse / delta arrive by LDS-DMA into a two-deep ring;
// the group sum happens in the accumulators (no partials, no atomics)
// ================================================================================================
__global__ __launch_bounds__(256) void dkdv_kernel(Args a) {
    constexpr int KS = 4, NF = 8, QT = 64;
    constexpr int STAT = 512;                                // lse[64] | delta[64] fp32 per ring slot
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // Q ring [2] | dO ring [2] | stats [2]  (65 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int group = a.Hq / a.Hkv;
    int xb, hk, b;
#ifndef MM355_DKDV_INNER
#define MM355_DKDV_INNER inner_heads(a.Hkv, 1)
#endif
    block_coords((a.L + 63) / 64, a.Hkv, MM355_DKDV_INNER, false, xb, hk, b);   // one block per KV head: it walks the query heads of its group
    const int kv0 = xb * 64;
    const int L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    const int mykey0 = kv0 + wave * 16;

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
        for (int v = lane; v < 16 * (DP / 8); v += 64) {
            const int r = v / (DP / 8), c = (v % (DP / 8)) * 8;
            const int key = mykey0 + r;
            if (key >= L) continue;
            f32x4 x0 = f32x4{0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (!zero) { x0 = *(const f32x4*)(so + r * DP + c); x1 = *(const f32x4*)(so + r * DP + c + 4); }
            const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            *(u32x4*)((is_dv ? a.dv : a.dk) + (row_base + key) * a.ld_dkv + (int64_t)hk * DP + c) = pack8(f);
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
    const int q_start = a.causal ? kv0 : 0;                  // kv0 is a multiple of 64 = QT
    const int n_it = seqlen > q_start ? (seqlen - q_start + QT - 1) / QT : 0;
    const int n_tot = n_it * group;                          // flattened (query head of the group, query tile) walk
    const uint16_t* qb = a.q + row_base * a.ld_q + (int64_t)hk * group * DP;
    const uint16_t* dob = a.d_o + row_base * a.ld_o + (int64_t)hk * group * DP;
    const float* lse_b = a.lse_in + ((int64_t)b * a.Hq + hk * group) * L;
    const float* del_b = a.delta + ((int64_t)b * a.Hq + hk * group) * L;
    TileSrc tq, td;
    tq.init(wave, lane, a.ld_q);
    td.init(wave, lane, a.ld_o);
    auto fetch = [&](int u) {                                // Q / dO / stats of walk step u -> ring slot u & 1
        const int g = u / n_it, qt0 = q_start + (u - g * n_it) * QT, slot = u & 1;
        dma_tile(qb + g * DP, a.ld_q, qt0, L, tq, smem + slot * TILE, wave, lane);
        dma_tile(dob + g * DP, a.ld_o, qt0, L, td, smem + (2 + slot) * TILE, wave, lane);
        if (wave < 2) {                                      // wave 0: lse, wave 1: delta (64 floats = one 256-B piece each)
            const float* sp = (wave == 0 ? lse_b : del_b) + (int64_t)g * L + min(qt0 + lane, L - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)(smem + 4 * TILE + slot * STAT + wave * 256), 4, 0, 0);
        }
    };
    if (n_tot > 0) fetch(0);

    // K / V fragments of this wave's 16 keys (B operands): lane holds X[key = fr][kk*32 + fq*8 ..]
    bf16x8 kf[KS], vf[KS];
    {
        const int key = min(mykey0 + fr, L - 1);
        const uint16_t* kp = a.k + (row_base + key) * a.ld_k + (int64_t)hk * DP;
        const uint16_t* vp = a.v + (row_base + key) * a.ld_k + (int64_t)hk * DP;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            kf[kk] = *(const bf16x8*)(kp + kk * 32 + fq * 8);
            vf[kk] = *(const bf16x8*)(vp + kk * 32 + fq * 8);
        }
    }
    int k_off[4];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) k_off[kk] = offN<DS>(fr, kk * 4 + fq);
    int t_off[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) t_off[j] = nat_perm_off<DS>(j, fr, fq);
    const float sl2 = a.scale * LOG2E;
    const f32x4 sl2v = f32x4{sl2, sl2, sl2, sl2}, scv = f32x4{a.scale, a.scale, a.scale, a.scale};
    const int kg = mykey0 + fr;

    for (int it = 0, itq = 0; it < n_tot; ++it, itq = (itq + 1 == n_it ? 0 : itq + 1)) {
        const int qt0 = q_start + itq * QT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 1 < n_tot) fetch(it + 1);
        const unsigned char* sQ = smem + (it & 1) * TILE;
        const unsigned char* sDO = smem + (2 + (it & 1)) * TILE;
        const float* sStat = (const float*)(smem + 4 * TILE + (it & 1) * STAT);

        // Two halves of 32 query rows (= one k-slot of the second pair of products).  Source order puts the score MFMAs of
        // half 1 between the softmax of half 0 and the dV / dK MFMAs of half 0, so that the matrix pipe has independent
        // work while the VALU does the exponentials (the whole body is one basic block per mask variant).
        const bool need_mask = (qt0 + QT > seqlen) || (kv0 + 64 > seqlen) || (a.causal && qt0 < kv0 + 64);
        auto body = [&](auto mask_c) {
            constexpr bool MASK = decltype(mask_c)::value;
            // visible query rows of this lane's key: [qmin, seqlen) -> tile-relative window [lo, lo + span)
            const int qmin = kg >= seqlen ? seqlen : (a.causal ? kg : 0);
            const int lo = qmin - qt0 - fq * 4;
            const unsigned span = (unsigned)(seqlen - qmin);
            f32x4 s[4], dp[4];                               // S / dP fragments: lane holds X[q = i*16 + fq*4 + r][key = fr]
            auto scores2 = [&](int i0) {                      // fragments i0, i0 + 1: four independent accumulator chains
#pragma unroll
                for (int i = i0; i < i0 + 2; ++i) { s[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                    for (int i = i0; i < i0 + 2; ++i) {
                        const bf16x8 qa = *(const bf16x8*)(sQ + i * 4096 + k_off[kk]);
                        const bf16x8 da = *(const bf16x8*)(sDO + i * 4096 + k_off[kk]);
                        s[i] = mfma16(qa, kf[kk], s[i]);
                        dp[i] = mfma16(da, vf[kk], dp[i]);
                    }
            };
            auto soft = [&](int i) {                         // P in place of S, dS = P o (dP - delta) * scale in place of dP
                const f32x4 nl4 = *(const f32x4*)(sStat + i * 16 + fq * 4) * -LOG2E;           // packed fp32 throughout
                const f32x4 nd4 = *(const f32x4*)(sStat + QT + i * 16 + fq * 4) * -a.scale;
                f32x4 p = __builtin_elementwise_fma(s[i], sl2v, nl4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(p[r]);
                    if (MASK) p[r] = ((unsigned)(i * 16 + r - lo) < span) ? p[r] : 0.f;
                }
                s[i] = p;
                dp[i] = p * __builtin_elementwise_fma(dp[i], scv, nd4);
            };
            auto grads = [&](int ks) {
                const bf16x8 pa = pack_acc(s[2 * ks], s[2 * ks + 1]);
                const bf16x8 dsa = pack_acc(dp[2 * ks], dp[2 * ks + 1]);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const bf16x8 dob8 = read_nat_perm_at<DS>(sDO, ks * 32, t_off[j]);
                    const bf16x8 qb8 = read_nat_perm_at<DS>(sQ, ks * 32, t_off[j]);
                    dvacc[j] = mfma16(pa, dob8, dvacc[j]);
                    dkacc[j] = mfma16(dsa, qb8, dkacc[j]);
                }
            };
            scores2(0);
            soft(0); soft(1);
            scores2(2);
            grads(0);
            soft(2); soft(3);
            grads(1);
        };
        if (need_mask) body(std::true_type{}); else body(std::false_type{});
    }
    __syncthreads();
    store_rows(dkacc, false, false);
    store_rows(dvacc, true, false);
}

}  // namespace attn3

static int fwd_pp_launch(const attn2::Args& a, hipStream_t s) {
    constexpr int LDS = 8 * attn3::TILE;                     // 128 KiB
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)attn3::fwd_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return MM355_ELAUNCH;
        attr_done = true;
    }
    const int64_t nblk = (int64_t)((a.L + 255) / 256) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(attn3::fwd_pp_kernel, dim3((unsigned)nblk), dim3(512), LDS, s, a);
    return mm_launch_status();
}

// RQ = 4 instantiations (one wave per SIMD, accumulators in AGPRs): attn3_wide.hip
int mm355_attn3_fwd_wide_launch(const attn2::Args& a, hipStream_t s);
int mm355_attn3_dq_wide_launch(const attn2::Args& a, hipStream_t s);

int mm355_attn3_fwd_launch(const attn2::Args& a, hipStream_t s) {
    static const bool pp = [] { const char* e = std::getenv("MM355_ATTN_PP"); return e && e[0] == '1'; }();
    if (pp) return fwd_pp_launch(a, s);
    // MM355_ATTN_RQ=4: the 64-rows-per-wave experiment (attn3_wide.hip); default: 32 rows per wave, two workgroups per CU
    static const int rq = [] { const char* e = std::getenv("MM355_ATTN_RQ"); return (e && e[0] == '4') ? 4 : 2; }();
    if (rq == 4) return mm355_attn3_fwd_wide_launch(a, s);
    const int64_t nblk = (int64_t)((a.L + 127) / 128) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(attn3::fwd_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return mm_launch_status();
}

int mm355_attn3_dq_launch(const attn2::Args& a, hipStream_t s) {
    static const int rq = [] { const char* e = std::getenv("MM355_ATTN_RQ_DQ"); return (e && e[0] == '4') ? 4 : 2; }();
    if (rq == 4) return mm355_attn3_dq_wide_launch(a, s);
    const int64_t nblk = (int64_t)((a.L + 127) / 128) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(attn3::dq_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return mm_launch_status();
}

int mm355_attn3_dkdv_launch(const attn2::Args& a, hipStream_t s) {
    constexpr int LDS = 4 * attn3::TILE + 2 * 512;           // 65 KiB > the default cap: raise it once
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)attn3::dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return MM355_ELAUNCH;
        attr_done = true;
    }
    const int64_t nblk = (int64_t)((a.L + 63) / 64) * a.Hkv * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    dim3 grid((unsigned)nblk);
    hipLaunchKernelGGL(attn3::dkdv_kernel, grid, dim3(256), LDS, s, a);
    return mm_launch_status();
}
