// Attention backward, d == 128, fourth generation (gfx950): the dK / dV kernel and the dQ kernel as hand-placed one-wave-per-SIMD
// instruction streams (tools/gen_attn4_bwd.py writes csrc/attn4_bwd_gen/*.inc; read its header and that of tools/gen_attn4.py first).
//   * products on v_mfma_f32_32x32x16_bf16; the side a wave owns for its whole life sits in the accumulator file as B operands (dK / dV:
//     K and V of 32 keys, as stored; dQ: Q and dO of 32 query rows), the other side streams through a four-slot LDS ring
//     of 64-row tiles (LDS-DMA, rows beyond the sample out of range of the descriptor = zeros); fragments go LDS -> accumulator registers
//     (ds_read_b128 for the score products, ds_read_b64_tr_b16 pairs for the gradient products, one tile swizzle serves both: 16-B chunk ^
//     ((row & 3) << 2 | (row >> 2) & 3))
//   * the score chains start from C = -lse / scale (units of the raw dot product) and C = -delta; the softmax scale enters on the fp32 side,
//     P = exp2((scale * log2 e) * x): per score one v_mul_f32, one v_exp_f32 and one v_mul_f32 (dS = P * y)
//     (dK / dV: the C values are read from the tile's statistics rows straight into the chain's registers; dQ: two constant tuples).
//     q and k enter the MFMAs as stored -- all three attention kernels recompute the SAME fp32 scores (rounds 1-4: a re-rounded bf16 copy of
//     one operand times scale * log2 e, a different one per kernel)
//   * dK / dV: workgroup = 128 keys of one KV head (4 waves x 32), walks the 64-row query tiles of all query heads of its GQA group in one
//     software pipeline (the group sum happens in the accumulators: no partials, no atomics); dQ: workgroup = 128 query rows (4 x 32)
//   * scale is applied once to the finished accumulators; the inverse RoPE rotation of dq / dk (mm355_attn_bwd_rope) in the epilogues
// Replaces the backward of torch SDPA as driven by HF LlamaModel (reference call site metamorph_llama.py:349-359).
#include "attn3_kernels.h"

#ifndef ATTN4B_GEN_DIR
#define ATTN4B_GEN_DIR attn4_bwd_gen
#endif
#define ATTN4B_STR2(x) #x
#define ATTN4B_STR(x) ATTN4B_STR2(x)
#define ATTN4B_INC(f) ATTN4B_STR(ATTN4B_GEN_DIR/f)

namespace attn4b {
using namespace attn2;
using attn3::block_coords;
using attn3::inner_heads;
using attn3::lptr_t;

constexpr float LOG2E = 1.4426950408889634f;
constexpr int SLOT = 32768;                                  // kv: Q tile | dO tile;  q: K tile | V tile
constexpr int KV_BASE = 2048;                                // kv: the four 512-B statistics slots come first
constexpr int KV_LDS = KV_BASE + 4 * SLOT, Q_LDS = 4 * SLOT;

MM_DEV int rot4(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// lane constants of the LDS reads (see the generator): RA b128 rows, TA / TB the two transposed gathers of a 16-row step, SA statistics
struct LaneAddr {
    int RA[8], TA[4], TB[4], SA;
    MM_DEV void init(int lane) {
        const int c = lane & 31, hi = lane >> 5, g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) RA[ks] = c * 256 + (((ks * 2 + hi) ^ rot4(c)) << 4);
        const int row0 = 4 * hi + (i >> 2), row1 = row0 + 8;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int chunk = db * 4 + (g & 1) * 2 + ((i & 3) >> 1);
            TA[db] = row0 * 256 + ((chunk ^ rot4(row0)) << 4) + (i & 1) * 8;
            TB[db] = row1 * 256 + ((chunk ^ rot4(row1)) << 4) + (i & 1) * 8;
        }
        SA = hi * 16;
    }
};
// source offsets of the four 1-KiB pieces a wave moves per 64-row tile: piece i = tile rows 16*wave + 4*i + (lane >> 4)
MM_DEV void piece_offsets(int wave, int lane, uint32_t ldb, uint32_t (&vo)[4]) {
    const int r4 = lane >> 4, pc = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) vo[i] = (uint32_t)(16 * wave + 4 * i + r4) * ldb + (uint32_t)((pc ^ ((r4 << 2) | i)) << 4);
}
MM_DEV uint32_t scale_pair(uint32_t w, float s) { return pack2bf(bflo(w) * s, bfhi(w) * s); }

#define BWD_BARRIER() asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NDMA) : "memory")

// ================================================================================================ dK / dV
template <bool SAFE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(96))) void dkdv_kernel(Args a, const float* __restrict__ nstat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("" ::: "v255", "a255");
#ifdef MM355_ATTN4_TIMING                                    // TIMING-ONLY build (tools/): phase stamps of wave 0 overwrite the block's first dk row
    const long long tm0 = __builtin_readcyclecounter();
    long long tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0;
#define BWD_STAMP(x) x = __builtin_readcyclecounter()
#else
#define BWD_STAMP(x) do {} while (0)
#endif
    constexpr int NDMA = 9;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hi = lane >> 5;
    const int rep = a.Hq / a.Hkv;
    int xb, hk, b;
    block_coords((a.L + 127) / 128, a.Hkv, inner_heads(a.Hkv, 1), false, xb, hk, b);
    const int kv0 = xb * 128;
    const int L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    const int kw0 = kv0 + 32 * wave;

    if (kv0 >= seqlen) {                                     // keys are all padding: zero gradients
        for (int v = tid; v < 128 * 16; v += 256) {
            const int r = v >> 4, cc = (v & 15) * 8;
            if (kv0 + r < L) {
                *(u32x4*)(a.dk + (row_base + kv0 + r) * a.ld_dkv + (int64_t)hk * 128 + cc) = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(a.dv + (row_base + kv0 + r) * a.ld_dkv + (int64_t)hk * 128 + cc) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        return;
    }
    const int q_start = a.causal ? kv0 : 0;                  // a multiple of 64
    const int n_it = (seqlen - q_start + 63) >> 6;           // query tiles per head (>= 1)
    const int n_tot = n_it * rep;
    const uint32_t ldqb = (uint32_t)a.ld_q * 2u, ldob = (uint32_t)a.ld_o * 2u;
    const int64_t stat_n = (int64_t)a.B * a.Hq * L;

    const uint32_t sl2b_ = __builtin_amdgcn_readfirstlane(__float_as_uint(a.scale * LOG2E));   // c = scale * log2 e: the SGPR operand of the v_mul_f32s
    LaneAddr la;
    la.init(lane);
    // ring slots 0, 1 are reached from RA / TA / TB (+ immediates < 64 KiB), slots 2, 3 from the copies 64 KiB higher
    int RA[8], TA[4], TB[4], RAH[8], TAH[4], TBH[4], SA = la.SA;
#pragma unroll
    for (int i = 0; i < 8; ++i) { RA[i] = la.RA[i] + KV_BASE; RAH[i] = RA[i] + 65536; asm volatile("" : "+v"(RA[i])); asm volatile("" : "+v"(RAH[i])); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        TA[i] = la.TA[i] + KV_BASE; TB[i] = la.TB[i] + KV_BASE; TAH[i] = TA[i] + 65536; TBH[i] = TB[i] + 65536;
        asm volatile("" : "+v"(TA[i])); asm volatile("" : "+v"(TB[i])); asm volatile("" : "+v"(TAH[i])); asm volatile("" : "+v"(TBH[i]));
    }
    asm volatile("" : "+v"(SA));
    uint32_t qvo[4], ovo[4], svo = (uint32_t)lane * 4u;
    piece_offsets(wave, lane, ldqb, qvo);
    piece_offsets(wave, lane, ldob, ovo);
#pragma unroll
    for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(qvo[i])); asm volatile("" : "+v"(ovo[i])); }
    asm volatile("" : "+v"(svo));
    const float ninf = -INFINITY;
    float t0_;
    // mask of this lane's key kw0 + c: visible query rows [qmin, seqlen)
    const int kg = kw0 + c;
    const int qmin = kg >= seqlen ? seqlen : (a.causal ? kg : 0);
    const int mlo_base = qmin - 4 * hi;
    const unsigned mspan_ = (unsigned)(seqlen - qmin);
    int mlo_ = 0;
    bool mask_cur = false;

    // tile walk: tile t = (query head g of the group, query tile it); two cursors: the tile being moved (d*) and the tile being computed (c*)
    __amdgpu_buffer_rsrc_t rsQ, rsO, rsS;
    int dst0 = 0;
    int dg = 0, dit = 0, cg = 0, cit = 0;
#define KV_TILE_VARS(slot_) do { \
        const bool live_ = dg < rep; \
        const int qt0_ = q_start + dit * 64; \
        const int hq_ = hk * rep + dg; \
        rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(a.q + (row_base + qt0_) * a.ld_q + (int64_t)hq_ * 128), 0, live_ ? (uint32_t)(seqlen - qt0_ - 1) * ldqb + 256u : 0u, 0x00020000); \
        rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.d_o + (row_base + qt0_) * a.ld_o + (int64_t)hq_ * 128), 0, live_ ? (uint32_t)(seqlen - qt0_ - 1) * ldob + 256u : 0u, 0x00020000); \
        rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(nstat + (wave & 1) * stat_n + ((int64_t)b * a.Hq + hq_) * L + qt0_), 0, live_ ? (uint32_t)(L - qt0_) * 4u : 0u, 0x00020000); \
        dst0 = (slot_); \
        if (++dit == n_it) { dit = 0; ++dg; } } while (0)
#define BWD_DMA(i) do { __builtin_amdgcn_sched_barrier(0); \
        if ((i) < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + KV_BASE + dst0 * SLOT + (wave * 4 + (i)) * 1024), 16, qvo[(i) & 3], 0, 0, 0); \
        else if ((i) < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + KV_BASE + dst0 * SLOT + 16384 + (wave * 4 + (i) - 4) * 1024), 16, ovo[(i) & 3], 0, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lptr_t)(smem + dst0 * 512 + (wave & 1) * 256), 4, svo, 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define KV_MASK_VARS() do { \
        const int qt0_ = q_start + cit * 64; \
        mask_cur = (a.causal && qt0_ < kw0 + 31) || qt0_ + 64 > seqlen || kw0 + 32 > seqlen; \
        mlo_ = mlo_base - qt0_; \
        if (++cit == n_it) { cit = 0; ++cg; } } while (0)

    KV_TILE_VARS(0);
#pragma unroll
    for (int i = 0; i < 9; ++i) { BWD_DMA(i); }
    KV_TILE_VARS(1);
#pragma unroll
    for (int i = 0; i < 9; ++i) { BWD_DMA(i); }
    // (the tiles are in flight: now the rows this wave keeps)  persistent operands: lane (key c, hi) holds X[kw0 + c][ks*16 + hi*8 .. + 8]
    {
        const int key = min(kw0 + c, L - 1);
        const uint16_t* p0_ = a.k + (row_base + key) * a.ld_k + (int64_t)hk * 128 + hi * 8;
        const uint16_t* p1_ = a.v + (row_base + key) * a.ld_k + (int64_t)hk * 128 + hi * 8;
#include ATTN4B_INC(kv_pers_load.inc)
    }
#include ATTN4B_INC(kv_zero.inc)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        float t1_;
#include ATTN4B_INC(kv_pers_place.inc)
    }
    __builtin_amdgcn_s_barrier();

    BWD_STAMP(tm1);
    KV_TILE_VARS(2);                                         // the head moves tile 2
    if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_head.inc)
    } else {
#include ATTN4B_INC(kv_head.inc)
    }
    BWD_STAMP(tm2);
    int u = 0;
    for (;;) {
        if (u >= n_tot - 1) break;
        KV_TILE_VARS(3); KV_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_loop0.inc)
        } else {
#include ATTN4B_INC(kv_loop0.inc)
        }
        ++u;
        if (u >= n_tot - 1) break;
        KV_TILE_VARS(0); KV_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_loop1.inc)
        } else {
#include ATTN4B_INC(kv_loop1.inc)
        }
        ++u;
        if (u >= n_tot - 1) break;
        KV_TILE_VARS(1); KV_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_loop2.inc)
        } else {
#include ATTN4B_INC(kv_loop2.inc)
        }
        ++u;
        if (u >= n_tot - 1) break;
        KV_TILE_VARS(2); KV_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_loop3.inc)
        } else {
#include ATTN4B_INC(kv_loop3.inc)
        }
        ++u;
    }
    BWD_STAMP(tm3);
    KV_MASK_VARS();
    switch (u & 3) {
    case 0:
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_tail0.inc)
        } else {
#include ATTN4B_INC(kv_tail0.inc)
        }
        break;
    case 1:
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_tail1.inc)
        } else {
#include ATTN4B_INC(kv_tail1.inc)
        }
        break;
    case 2:
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_tail2.inc)
        } else {
#include ATTN4B_INC(kv_tail2.inc)
        }
        break;
    default:
        if constexpr (SAFE) {
#include ATTN4B_INC(kv_safe_tail3.inc)
        } else {
#include ATTN4B_INC(kv_tail3.inc)
        }
        break;
    }
#undef BWD_DMA
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                         // the ring becomes the output staging area
    BWD_STAMP(tm4);

    // epilogue: lane (key c, hi) holds X^T[d = db*32 + 8*(r >> 2) + 4*hi + (r & 3)][key]; dK *= scale, optional inverse RoPE on the
    // bf16-rounded dK (partner column d + 64 = block db + 2 of the same lane), bf16 [key][d] through this wave's 8 KiB -> row stores
    unsigned char* so = smem + wave * 16384;
    const int pos = min(kg, L - 1) + (a.rope_pos ? a.rope_pos[b] : 0);
#pragma unroll
    for (int which_ = 0; which_ < 2; ++which_) {             // 0: dK, 1: dV
        const float sc = which_ == 0 ? a.scale : 1.0f;
#pragma unroll
        for (int dbp = 0; dbp < 2; ++dbp)
#pragma unroll
            for (int i4_ = 0; i4_ < 4; ++i4_) {
                float x_[4], lo_[4], hi_[4];
                {
                    const int db_ = dbp;
#include ATTN4B_INC(kv_acc_read.inc)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lo_[r] = x_[r] * sc;
                }
                {
                    const int db_ = dbp + 2;
#include ATTN4B_INC(kv_acc_read.inc)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hi_[r] = x_[r] * sc;
                }
                if (which_ == 0 && a.rope_cos) {
                    const int col = dbp * 32 + 8 * i4_ + 4 * hi;
                    const u32x2 cw = *(const u32x2*)(a.rope_cos + (int64_t)pos * 128 + col);
                    const u32x2 sw = *(const u32x2*)(a.rope_sin + (int64_t)pos * 128 + col);
                    const float cs[4] = {bflo(cw.x), bfhi(cw.x), bflo(cw.y), bfhi(cw.y)};
                    const float sn[4] = {bflo(sw.x), bfhi(sw.x), bflo(sw.y), bfhi(sw.y)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x1 = round_bf(lo_[r]), x2 = round_bf(hi_[r]);
                        lo_[r] = x1 * cs[r] + x2 * sn[r];
                        hi_[r] = x2 * cs[r] - x1 * sn[r];
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int db = dbp + 2 * half;
                    const float* s4 = half ? hi_ : lo_;
                    u32x2 w;
                    w.x = pack2bf(s4[0], s4[1]);
                    w.y = pack2bf(s4[2], s4[3]);
                    *(u32x2*)(so + which_ * 8192 + c * 256 + (((db * 4 + i4_) ^ (c & 15)) << 4) + hi * 8) = w;
                }
            }
    }
    __syncthreads();
#pragma unroll
    for (int which_ = 0; which_ < 2; ++which_)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = 4 * k + (lane >> 4), j = lane & 15;
            const int key = kw0 + r;
            if (key < L) {
                u32x4 val = *(const u32x4*)(so + which_ * 8192 + r * 256 + ((j ^ (r & 15)) << 4));
                if (key >= seqlen) val = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)((which_ ? a.dv : a.dk) + (row_base + key) * a.ld_dkv + (int64_t)hk * 128 + j * 8) = val;
            }
        }
#ifdef MM355_ATTN4_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tm5 = __builtin_readcyclecounter();
    __syncthreads();
    if (tid == 0) {
        long long* w = (long long*)(a.dk + (row_base + kv0) * a.ld_dkv + (int64_t)hk * 128);
        w[0] = tm1 - tm0; w[1] = tm2 - tm1; w[2] = tm3 - tm2; w[3] = tm4 - tm3; w[4] = tm5 - tm4; w[5] = n_tot;
    }
#endif
}

// ================================================================================================ dQ
template <bool SAFE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(80))) void dq_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("" ::: "v255", "a255");
    constexpr int NDMA = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hi = lane >> 5;
    int xb, hq, b;
    block_coords((a.L + 127) / 128, a.Hq, inner_heads(a.Hq, a.Hq / a.Hkv), true, xb, hq, b);
    const int q0 = xb * 128;
    const int hk = hq / (a.Hq / a.Hkv);
    const int L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    uint16_t* dq_base = a.dqb + row_base * a.ld_dq + (int64_t)hq * 128;
    if (q0 >= seqlen) {                                      // padded query rows carry zero gradient
        for (int v = tid; v < 128 * 16; v += 256) {
            const int r = v >> 4, cc = (v & 15) * 8;
            if (q0 + r < L) *(u32x4*)(dq_base + (int64_t)(q0 + r) * a.ld_dq + cc) = u32x4{0u, 0u, 0u, 0u};
        }
        return;
    }
    const int kv_end = a.causal ? min(seqlen, q0 + 128) : seqlen;
    const int n_tot = (kv_end + 63) >> 6;
    const int qw0 = q0 + 32 * wave;
    const int qg = qw0 + c;
    const uint32_t ldb = (uint32_t)a.ld_k * 2u;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * 128;
    const uint16_t* vbase = a.v + row_base * a.ld_k + (int64_t)hk * 128;
    const uint32_t nrec = (uint32_t)(seqlen - 1) * ldb + 256u;

    const uint32_t sl2b_ = __builtin_amdgcn_readfirstlane(__float_as_uint(a.scale * LOG2E));   // c = scale * log2 e: the SGPR operand of the v_mul_f32s
    LaneAddr la;
    la.init(lane);
    int RA[8], TA[4], TB[4], RAH[8], TAH[4], TBH[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { RA[i] = la.RA[i]; RAH[i] = RA[i] + 65536; asm volatile("" : "+v"(RA[i])); asm volatile("" : "+v"(RAH[i])); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        TA[i] = la.TA[i]; TB[i] = la.TB[i]; TAH[i] = TA[i] + 65536; TBH[i] = TB[i] + 65536;
        asm volatile("" : "+v"(TA[i])); asm volatile("" : "+v"(TB[i])); asm volatile("" : "+v"(TAH[i])); asm volatile("" : "+v"(TBH[i]));
    }
    uint32_t kvo[4];
    piece_offsets(wave, lane, ldb, kvo);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(kvo[i]));
    const float ninf = -INFINITY;
    const int kmax = qg >= seqlen ? -1 : (a.causal ? min(qg, seqlen - 1) : seqlen - 1);      // last visible key of this lane's query row
    const int mlim_base = kmax - 4 * hi;
    int mlim_ = 0;
    bool mask_cur = false;

    __amdgpu_buffer_rsrc_t rsK, rsV;
    int dst0 = 0, dt = 0, ct = 0;
#define Q_TILE_VARS(slot_) do { \
        rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(kbase + (int64_t)dt * 64 * a.ld_k), 0, dt < n_tot ? nrec - (uint32_t)dt * 64u * ldb : 0u, 0x00020000); \
        rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(vbase + (int64_t)dt * 64 * a.ld_k), 0, dt < n_tot ? nrec - (uint32_t)dt * 64u * ldb : 0u, 0x00020000); \
        dst0 = (slot_) * SLOT; ++dt; } while (0)
#define BWD_DMA(i) do { __builtin_amdgcn_sched_barrier(0); \
        if ((i) < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + dst0 + (wave * 4 + (i)) * 1024), 16, kvo[(i) & 3], 0, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + dst0 + 16384 + (wave * 4 + (i) - 4) * 1024), 16, kvo[(i) & 3], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define Q_MASK_VARS() do { \
        const int kt0_ = ct * 64; \
        mask_cur = (a.causal && kt0_ + 63 > qw0) || kt0_ + 64 > seqlen || qw0 + 32 > seqlen; \
        mlim_ = mlim_base - kt0_; ++ct; } while (0)

    Q_TILE_VARS(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) { BWD_DMA(i); }
    Q_TILE_VARS(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { BWD_DMA(i); }
    // (the tiles are in flight: now the rows this wave keeps)
    {
        const int qc = min(qg, L - 1);
        const uint16_t* p0_ = a.q + (row_base + qc) * a.ld_q + (int64_t)hq * 128 + hi * 8;
        const uint16_t* p1_ = a.d_o + (row_base + qc) * a.ld_o + (int64_t)hq * 128 + hi * 8;
#include ATTN4B_INC(q_pers_load.inc)
        const float nl_ = -a.lse_in[((int64_t)b * a.Hq + hq) * L + qc] * (1.0f / a.scale);
        const float nd_ = -a.delta[((int64_t)b * a.Hq + hq) * L + qc];
#include ATTN4B_INC(q_tuple_write.inc)
    }
#include ATTN4B_INC(q_zero.inc)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        float t0_, t1_;
#include ATTN4B_INC(q_pers_place.inc)
    }
    __builtin_amdgcn_s_barrier();

    Q_TILE_VARS(2);
    if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_head.inc)
    } else {
#include ATTN4B_INC(q_head.inc)
    }
    int u = 0;
    for (;;) {
        if (u >= n_tot - 1) break;
        Q_TILE_VARS(3); Q_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_loop0.inc)
        } else {
#include ATTN4B_INC(q_loop0.inc)
        }
        ++u;
        if (u >= n_tot - 1) break;
        Q_TILE_VARS(0); Q_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_loop1.inc)
        } else {
#include ATTN4B_INC(q_loop1.inc)
        }
        ++u;
        if (u >= n_tot - 1) break;
        Q_TILE_VARS(1); Q_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_loop2.inc)
        } else {
#include ATTN4B_INC(q_loop2.inc)
        }
        ++u;
        if (u >= n_tot - 1) break;
        Q_TILE_VARS(2); Q_MASK_VARS();
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_loop3.inc)
        } else {
#include ATTN4B_INC(q_loop3.inc)
        }
        ++u;
    }
    Q_MASK_VARS();
    switch (u & 3) {
    case 0:
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_tail0.inc)
        } else {
#include ATTN4B_INC(q_tail0.inc)
        }
        break;
    case 1:
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_tail1.inc)
        } else {
#include ATTN4B_INC(q_tail1.inc)
        }
        break;
    case 2:
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_tail2.inc)
        } else {
#include ATTN4B_INC(q_tail2.inc)
        }
        break;
    default:
        if constexpr (SAFE) {
#include ATTN4B_INC(q_safe_tail3.inc)
        } else {
#include ATTN4B_INC(q_tail3.inc)
        }
        break;
    }
#undef BWD_DMA
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // epilogue: lane (q c, hi) holds dQ^T[d = db*32 + 8*(r >> 2) + 4*hi + (r & 3)][q]; * scale, optional inverse RoPE, zero for padded rows
    unsigned char* so = smem + wave * 8192;
    const float keep = qg < seqlen ? a.scale : 0.0f;
    const int pos = min(qg, L - 1) + (a.rope_pos ? a.rope_pos[b] : 0);
#pragma unroll
    for (int dbp = 0; dbp < 2; ++dbp)
#pragma unroll
        for (int i4_ = 0; i4_ < 4; ++i4_) {
            constexpr int which_ = 0;
            float x_[4], lo_[4], hi_[4];
            {
                const int db_ = dbp;
#include ATTN4B_INC(q_acc_read.inc)
#pragma unroll
                for (int r = 0; r < 4; ++r) lo_[r] = x_[r] * keep;
            }
            {
                const int db_ = dbp + 2;
#include ATTN4B_INC(q_acc_read.inc)
#pragma unroll
                for (int r = 0; r < 4; ++r) hi_[r] = x_[r] * keep;
            }
            if (a.rope_cos) {
                const int col = dbp * 32 + 8 * i4_ + 4 * hi;
                const u32x2 cw = *(const u32x2*)(a.rope_cos + (int64_t)pos * 128 + col);
                const u32x2 sw = *(const u32x2*)(a.rope_sin + (int64_t)pos * 128 + col);
                const float cs[4] = {bflo(cw.x), bfhi(cw.x), bflo(cw.y), bfhi(cw.y)};
                const float sn[4] = {bflo(sw.x), bfhi(sw.x), bflo(sw.y), bfhi(sw.y)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x1 = round_bf(lo_[r]), x2 = round_bf(hi_[r]);
                    lo_[r] = x1 * cs[r] + x2 * sn[r];
                    hi_[r] = x2 * cs[r] - x1 * sn[r];
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int db = dbp + 2 * half;
                const float* s4 = half ? hi_ : lo_;
                u32x2 w;
                w.x = pack2bf(s4[0], s4[1]);
                w.y = pack2bf(s4[2], s4[3]);
                *(u32x2*)(so + c * 256 + (((db * 4 + i4_) ^ (c & 15)) << 4) + hi * 8) = w;
            }
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = 4 * k + (lane >> 4), j = lane & 15;
        if (qw0 + r < L) *(u32x4*)(dq_base + (int64_t)(qw0 + r) * a.ld_dq + j * 8) = *(const u32x4*)(so + r * 256 + ((j ^ (r & 15)) << 4));
    }
}

// nstat[0][b][h][l] = -lse / scale (raw q . k units), nstat[1][b][h][l] = -delta: the C operands of the dK / dV kernel's score chains
__global__ __launch_bounds__(256) void nstat_kernel(const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ nstat, int64_t n,
                                                    float inv_scale) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        nstat[i] = -lse[i] * inv_scale;
        nstat[n + i] = -delta[i];
    }
}

}  // namespace attn4b

// variant 1 = the serialised debugging streams
int mm355_attn4_bwd_launch(const attn2::Args& a, float* workspace, int variant, hipStream_t s) {
    static std::atomic<uint64_t> ok_kv{0}, ok_kvs{0}, ok_q{0}, ok_qs{0};
    const int64_t n = (int64_t)a.B * a.Hq * a.L;
    hipLaunchKernelGGL(attn4b::nstat_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, a.lse_in, a.delta, workspace, n, 1.0f / a.scale);
    const int64_t nkv = (int64_t)((a.L + 127) / 128) * a.Hkv * a.B, nq = (int64_t)((a.L + 127) / 128) * a.Hq * a.B;
    if (nkv > 0x7fffffff || nq > 0x7fffffff) return MM355_EINVAL;
    if (variant == 1) {
        if (mm_ensure_dynamic_lds((const void*)attn4b::dkdv_kernel<true>, attn4b::KV_LDS, ok_kvs) != MM355_OK) return MM355_ELAUNCH;
        if (mm_ensure_dynamic_lds((const void*)attn4b::dq_kernel<true>, attn4b::Q_LDS, ok_qs) != MM355_OK) return MM355_ELAUNCH;
        hipLaunchKernelGGL(attn4b::dkdv_kernel<true>, dim3((unsigned)nkv), dim3(256), attn4b::KV_LDS, s, a, (const float*)workspace);
        hipLaunchKernelGGL(attn4b::dq_kernel<true>, dim3((unsigned)nq), dim3(256), attn4b::Q_LDS, s, a);
    } else {
        if (mm_ensure_dynamic_lds((const void*)attn4b::dkdv_kernel<false>, attn4b::KV_LDS, ok_kv) != MM355_OK) return MM355_ELAUNCH;
        if (mm_ensure_dynamic_lds((const void*)attn4b::dq_kernel<false>, attn4b::Q_LDS, ok_q) != MM355_OK) return MM355_ELAUNCH;
        hipLaunchKernelGGL(attn4b::dkdv_kernel<false>, dim3((unsigned)nkv), dim3(256), attn4b::KV_LDS, s, a, (const float*)workspace);
        hipLaunchKernelGGL(attn4b::dq_kernel<false>, dim3((unsigned)nq), dim3(256), attn4b::Q_LDS, s, a);
    }
    return mm_launch_status();
}
