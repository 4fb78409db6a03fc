// Attention, d == 128 fast path (gfx950): launchers + the dK/dV kernel.
// The forward / dQ kernel templates live in attn3_kernels.h.
#include "attn3_kernels.h"

namespace attn3 {
using namespace attn2;

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
    block_coords((a.L + 63) / 64, a.Hkv, inner_heads(a.Hkv, 1), false, xb, hk, b);   // one block per KV head: it walks the query heads of its group
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
            float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            if (a.rope_cos && !is_dv && !zero) {
                // inverse RoPE on the bf16-rounded dk (mm355_rope_qk(inverse)): the partner columns c ^ 64 of this key row are in the slab too
                const int cp = c ^ 64;
                const f32x4 p0 = *(const f32x4*)(so + r * DP + cp), p1 = *(const f32x4*)(so + r * DP + cp + 4);
                const float pf[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                const int pos = key + (a.rope_pos ? a.rope_pos[b] : 0);
                float cs[8], sn[8];
                unpack8(*(const u32x4*)(a.rope_cos + (int64_t)pos * DP + (c & 63)), cs);
                unpack8(*(const u32x4*)(a.rope_sin + (int64_t)pos * DP + (c & 63)), sn);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float own = round_bf(f[e]), par = round_bf(pf[e]);
                    f[e] = c < 64 ? own * cs[e] + par * sn[e] : own * cs[e] - par * sn[e];
                }
            }
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


int mm355_attn3_fwd_launch(const attn2::Args& a, hipStream_t s) {
    const int64_t nblk = (int64_t)((a.L + 127) / 128) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(attn3::fwd_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return mm_launch_status();
}

int mm355_attn3_dq_launch(const attn2::Args& a, hipStream_t s) {
    const int64_t nblk = (int64_t)((a.L + 127) / 128) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(attn3::dq_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return mm_launch_status();
}

int mm355_attn3_dkdv_launch(const attn2::Args& a, hipStream_t s) {
    constexpr int LDS = 4 * attn3::TILE + 2 * 512;           // 65 KiB > the default cap: raise it once
    static std::atomic<uint64_t> lds_ok{0};                  // per-device opt-in (mm355_common.h)
    if (mm_ensure_dynamic_lds((const void*)attn3::dkdv_kernel, LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    const int64_t nblk = (int64_t)((a.L + 63) / 64) * a.Hkv * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    dim3 grid((unsigned)nblk);
    hipLaunchKernelGGL(attn3::dkdv_kernel, grid, dim3(256), LDS, s, a);
    return mm_launch_status();
}
