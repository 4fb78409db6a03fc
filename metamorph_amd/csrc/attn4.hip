// Attention forward, d == 128, fourth generation (gfx950): ONE wave per SIMD with the whole 512-entry register file and a hand-placed
// instruction stream (tools/gen_attn4.py writes csrc/attn4_gen/*.inc; read its header first).
//   * workgroup = 256 query rows = 4 waves x 64 rows (two 32-row blocks per wave); products on v_mfma_f32_32x32x16_bf16:
//     S^T[64 keys][64 q] = K Q^T (32 MFMAs per tile), O^T[128 d][64 q] += V^T P^T (32 MFMAs per tile)
//   * the score MFMAs take q and k AS STORED (the bf16 values the reference's SDPA sees) and the first MFMA of every chain takes C = -m
//     (the running row maximum, in units of the raw dot product) from a register tuple: the chain delivers s' = q . k - m in fp32, and the
//     softmax scale enters on the fp32 side, P = exp2((scale * log2 e) * s'): one v_mul_f32 + one v_exp_f32 per score
//     (no subtract on the vector ALU); m only moves when some row grew by more than 2^6 (rare, wave-uniform branch).  Rounds 1-4 folded
//     scale * log2 e into a re-rounded bf16 copy of q: an extra rounding whose score error grows with |s|
//   * register file, all LITERAL registers owned by the asm stream (map in tools/gen_attn4.py): O a[0:127], Q a[128:191], K / V fragment
//     rings a[192:255] filled straight from LDS (ds_read_b128 / ds_read_b64_tr_b16 into accumulator registers); v[64:191] two score
//     tiles, v[192:223] the -m tuples, v[224:255] P; hipcc allocates only v[0:63] (amdgpu_num_vgpr(64): addresses, row sums, row maxima;
//     tools/audit_attn4.py proves on the emitted code object that no compiler instruction touches anything else)
//   * K / V tiles of 64 keys by LDS-DMA (buffer_load ... lds, rows beyond the sample's length are out of range of the descriptor and
//     arrive as zeros, so no row is ever clamped or read twice) into two-slot rings; ONE barrier per tile, placed after the tile's last
//     V read so that two slots suffice; K tile swizzle chunk ^ (row & 15) (conflict-free b128 reads of 32 rows x 32 B), V tile swizzle
//     64-B slot ^ (row & 3) (conflict-free transposed b64 reads of 4 rows x 64 B)
//   * software pipeline per wave: step t = { QK(t + 1) || exponentials + packing of tile t || V reads }  { PV(t) || row sums of tile t ||
//     row maxima of tile t + 1 || K reads || DMA of tiles t + 3 / t + 2 }; a wave's last tile (the only masked one) drains unpipelined
// Replaces torch SDPA as driven by HF LlamaModel (reference call site metamorph_llama.py:349-359).
#include "attn3_kernels.h"

// the generated streams; timing-only ablation builds (tools/gen_attn4.py --abl ...) compile with -DATTN4_GEN_DIR=attn4_gen_<name>
#ifndef ATTN4_GEN_DIR
#define ATTN4_GEN_DIR attn4_gen
#endif
#define ATTN4_STR2(x) #x
#define ATTN4_STR(x) ATTN4_STR2(x)
#define ATTN4_INC(f) ATTN4_STR(ATTN4_GEN_DIR/f)

namespace attn4 {
using namespace attn2;
using attn3::block_coords;
using attn3::inner_heads;
using attn3::lptr_t;

constexpr int TILE = 16384;                                  // [64 rows][128] bf16
constexpr int VRING = 2 * TILE;                              // K slots 0, 1 | V slots 0, 1
constexpr float LOG2E = 1.4426950408889634f;
constexpr float THR = 6.0f;                                  // log2 units: the running maximum stays while no row grew by more than 2^6

MM_DEV float half_swap_max(float m) {                        // max over the two lanes (l, l ^ 32) that share a query row
    float a = m, b = m;
    asm volatile("" : "+v"(b));
    swap32(a, b);
    return max2_raw(a, b);
}
MM_DEV float half_swap_sum(float m) {
    float a = m, b = m;
    asm volatile("" : "+v"(b));
    swap32(a, b);
    return a + b;
}

#define ATTN4_BARRIER() do { if (do_bar) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
// one 1-KiB piece of the tile the step's descriptor points at (base = the tile's first row, num_records = what is left of the sample from
// there, 0 for a tile nobody needs: every lane out of range, zeros, no memory access); the lane part of the address never changes
#define ATTN4_DMA_K(i) do { __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + kdst + (i) * 1024), 16, kvo[i], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ATTN4_DMA_V(i) do { if (do_bar) { __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + vdst + (i) * 1024), 16, vvo[i], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); } } while (0)
template <bool SAFE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(64))) void fwd_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];
    asm volatile("" ::: "v255", "a255");                   // the stream's literal registers: the descriptor must allocate the whole file
#ifdef MM355_ATTN4_TIMING                                    // TIMING-ONLY build (tools/): phase stamps of waves 0 and 3 overwrite the block's lse rows
    const long long tm0 = __builtin_readcyclecounter();
    long long tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0, tm5 = 0;
#define ATTN4_STAMP(x) x = __builtin_readcyclecounter()
#else
#define ATTN4_STAMP(x) do {} while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hi = lane >> 5;
    int xb, hq, b;
    block_coords((a.L + 255) / 256, a.Hq, inner_heads(a.Hq, a.Hq / a.Hkv), true, xb, hq, b);
    const int q0 = xb * 256;
    const int hk = hq / (a.Hq / a.Hkv);
    const int L = a.L;
    const int seqlen = a.seqlens ? min(a.seqlens[b], L) : L;
    const int64_t row_base = (int64_t)b * L;
    uint16_t* o_base = a.o + row_base * a.ld_o + (int64_t)hq * 128;
    float* lse_base = a.lse + ((int64_t)b * a.Hq + hq) * L;

    if (q0 >= seqlen) {                                      // whole block is padding: o = 0, lse = 0
        for (int v = tid; v < 256 * 16; v += 256) {
            const int r = v >> 4, cc = (v & 15) * 8;
            if (q0 + r < L) *(u32x4*)(o_base + (int64_t)(q0 + r) * a.ld_o + cc) = u32x4{0u, 0u, 0u, 0u};
        }
        if (q0 + tid < L) lse_base[q0 + tid] = 0.f;
        return;
    }
    const int kv_end = a.causal ? min(seqlen, q0 + 256) : seqlen;
    const int T = (kv_end + 63) >> 6;                        // tiles the workgroup walks
    const int qw0 = q0 + wave * 64;
    // this wave's last tile (the only one that needs a mask); -1: all its rows are padding, it only moves tiles
    const int tw = qw0 < seqlen ? (a.causal ? min(T - 1, min(qw0 + 63, seqlen - 1) >> 6) : T - 1) : -1;

    const uint32_t ldb = (uint32_t)a.ld_k * 2u;
    const uint16_t* kbase = a.k + row_base * a.ld_k + (int64_t)hk * 128;
    const uint16_t* vbase = a.v + row_base * a.ld_k + (int64_t)hk * 128;
    const uint32_t nrec = (uint32_t)(seqlen - 1) * ldb + 256u;                             // rows >= seqlen are out of range: zeros
    __amdgpu_buffer_rsrc_t rsK, rsV;
#define ATTN4_TILE_RSRC(base_, tile_) __builtin_amdgcn_make_buffer_rsrc((void*)((base_) + (int64_t)(tile_) * 64 * a.ld_k), 0, \
    (tile_) < T ? nrec - (uint32_t)(tile_) * 64u * ldb : 0u, 0x00020000)
    // piece i of this wave = tile rows 16*wave + 4*i + (lane >> 4); LDS side lane-linear, swizzle on the source chunk
    uint32_t kvo[4], vvo[4];
    {
        const int r4 = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t row = 16 * wave + 4 * i + r4;
            kvo[i] = row * ldb + (uint32_t)((pc ^ (4 * i + r4)) << 4);
            vvo[i] = row * ldb + (uint32_t)((pc ^ (r4 << 2)) << 4);
        }
    }
    // Q rows first (HBM latency is the prologue's critical path): lane holds Q[q = qb*32 + c][d = ks*16 + hi*8 .. + 8] -> v[128:191]
    const float sl2 = a.scale * LOG2E;                       // raw q . k units -> log2 domain; its bits in an SGPR: the scalar operand of the v_mul_f32s
    const uint32_t sl2b_ = __builtin_amdgcn_readfirstlane(__float_as_uint(sl2));
    {
        const uint16_t* qp0_ = a.q + (row_base + min(qw0 + c, L - 1)) * a.ld_q + (int64_t)hq * 128 + hi * 8;
        const uint16_t* qp1_ = a.q + (row_base + min(qw0 + 32 + c, L - 1)) * a.ld_q + (int64_t)hq * 128 + hi * 8;
#include ATTN4_INC(q_load.inc)
    }
    bool do_bar = true;
    int kdst, vdst;
    // K(0), V(0) -> slot 0, K(1) -> slot 1
    rsK = ATTN4_TILE_RSRC(kbase, 0); rsV = ATTN4_TILE_RSRC(vbase, 0); kdst = wave * 4096; vdst = VRING + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ATTN4_DMA_K(i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { ATTN4_DMA_V(i); }
    rsK = ATTN4_TILE_RSRC(kbase, 1); kdst = TILE + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ATTN4_DMA_K(i); }
#include ATTN4_INC(zero_o.inc)
    float LS[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // [qb][r & 3] partial row sums of this lane's keys
    float MX[2][2], rm_[2], rt_[2];
    uint64_t grow_ = 0;
    int nresc_ = 0;                                          // deferred-rescale branches this wave took (wave-uniform: lives in an SGPR)
#define ATTN4_COUNT_RESCALE() (++nresc_)
    float thr_ = THR / sl2;                                  // the same 2^6, in raw q . k units
    asm volatile("" : "+v"(thr_));
    const float ninf = -INFINITY;
    // LDS read addresses: K rows (b128), logical chunk ks*2 + hi of row c at physical chunk ^ (c & 15); V gathers (tr_b64): lane i of a
    // 16-lane group g supplies row 4*hi + (i >> 2), columns db*32 + (g & 1)*16 + (i & 3)*4 .. + 3, 64-B slot db ^ (row & 3)
    int KA[8], VA[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) KA[ks] = c * 256 + (((ks * 2 + hi) ^ (c & 15)) << 4);
    {
        const int g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int db = 0; db < 4; ++db)
            VA[db] = VRING + (4 * hi + (i >> 2)) * 256 + ((((db * 4 + (g & 1) * 2 + ((i & 3) >> 1)) ^ ((i >> 2) << 2))) << 4) + (i & 1) * 8;
    }
    // opaque from here on: hipcc would otherwise re-derive these lane constants in the middle of the stream (it owns 64 registers only)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(KA[ks]));
#pragma unroll
    for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(VA[i])); asm volatile("" : "+v"(kvo[i])); asm volatile("" : "+v"(vvo[i])); }
    int lim2[2] = {0, 0};
    if (tw >= 0) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qg = qw0 + qb * 32 + c;
            lim2[qb] = (a.causal ? min(qg, seqlen - 1) : seqlen - 1) - tw * 64 - 4 * hi;
        }
    }
    // Q fragments into the accumulator file: d-steps 0 and 1 here (the 16 row loads precede the 12 DMA pieces in the memory queue), 2..7 inside the head
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#include ATTN4_INC(q_pre01.inc)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

// step s moves K(s + 2) (phase A, before barrier s: its slot was K(s)'s, whose last reader passed barrier s - 1) and V(s + 2) (behind barrier s)
#define ATTN4_STEP_VARS(s_) do { const int s__ = (s_); \
    rsK = ATTN4_TILE_RSRC(kbase, s__ + 2); rsV = ATTN4_TILE_RSRC(vbase, s__ + 2); \
    kdst = (s__ & 1) * TILE + wave * 4096; vdst = VRING + (s__ & 1) * TILE + wave * 4096; } while (0)

    ATTN4_STAMP(tm1);
    if (tw >= 0) {
        bool mask_next = tw == 0;
        ATTN4_STEP_VARS(-1);
        if constexpr (SAFE) {
#include ATTN4_INC(safe_head.inc)
        } else {
#include ATTN4_INC(head.inc)
        }
        ATTN4_STAMP(tm2);
        int t = 0;
        for (;;) {
            if (t >= tw) break;
            mask_next = t + 1 == tw;
            ATTN4_STEP_VARS(t);
            if constexpr (SAFE) {
#include ATTN4_INC(safe_loop0.inc)
            } else {
#include ATTN4_INC(loop0.inc)
            }
            ++t;
            if (t >= tw) break;
            mask_next = t + 1 == tw;
            ATTN4_STEP_VARS(t);
            if constexpr (SAFE) {
#include ATTN4_INC(safe_loop1.inc)
            } else {
#include ATTN4_INC(loop1.inc)
            }
            ++t;
        }
        ATTN4_STAMP(tm3);
        do_bar = tw <= T - 2;
        ATTN4_STEP_VARS(tw);
        if (tw & 1) {
            if constexpr (SAFE) {
#include ATTN4_INC(safe_tail1.inc)
            } else {
#include ATTN4_INC(tail1.inc)
            }
        } else {
            if constexpr (SAFE) {
#include ATTN4_INC(safe_tail0.inc)
            } else {
#include ATTN4_INC(tail0.inc)
            }
        }
    }
    ATTN4_STAMP(tm4);
    // a wave that is done keeps moving its quarter of every remaining tile and meets the barriers of steps <= T - 2
    do_bar = true;
    for (int s = tw < 0 ? -1 : tw + 1; s <= T - 2; ++s) {
        ATTN4_STEP_VARS(s);
        if (s >= 0) {                                        // (K(1) left with the prologue)
#pragma unroll
            for (int i = 0; i < 4; ++i) { ATTN4_DMA_K(i); }
        }
        ATTN4_BARRIER();
#pragma unroll
        for (int i = 0; i < 4; ++i) { ATTN4_DMA_V(i); }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // last P V MFMAs, stray DMA
    __syncthreads();                                         // every wave is done with the rings: they become the output staging area
    ATTN4_STAMP(tm5);

    // epilogue: O = O^T / l -> bf16 [q][d] in this wave's 16 KiB (16-B chunk ^ (row & 15)) -> row-contiguous 16-B stores
    unsigned char* so = smem + wave * TILE;
#pragma unroll
    for (int qb_ = 0; qb_ < 2; ++qb_) {
        const int qg = qw0 + qb_ * 32 + c;
        const bool valid = qg < seqlen && tw >= 0;
        const float l_run = half_swap_sum((LS[qb_][0] + LS[qb_][1]) + (LS[qb_][2] + LS[qb_][3]));
        const float inv = (valid && l_run > 0.f) ? 1.0f / l_run : 0.f;
#pragma unroll
        for (int db_ = 0; db_ < 4; ++db_) {
            float x_[16];
#include ATTN4_INC(epi_read.inc)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                u32x2 w;
                w.x = pack2bf(x_[4 * i4 + 0] * inv, x_[4 * i4 + 1] * inv);
                w.y = pack2bf(x_[4 * i4 + 2] * inv, x_[4 * i4 + 3] * inv);
                *(u32x2*)(so + (qb_ * 32 + c) * 256 + (((db_ * 4 + i4) ^ (c & 15)) << 4) + hi * 8) = w;
            }
        }
        float nm0;
        if (qb_ == 0) asm volatile("v_mov_b32 %0, v192" : "=v"(nm0)); else asm volatile("v_mov_b32 %0, v208" : "=v"(nm0));
        if (hi == 0 && qg < L) lse_base[qg] = valid ? (log2f(l_run) - nm0 * sl2) * 0.6931471805599453f : 0.f;   // nm0 = -m in raw units
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = 4 * k + (lane >> 4), j = lane & 15;
        const int qg = qw0 + r;
        if (qg < L) *(u32x4*)(o_base + (int64_t)qg * a.ld_o + j * 8) = *(const u32x4*)(so + r * 256 + ((j ^ (r & 15)) << 4));
    }
    if (a.dbg && lane == 0) a.dbg[(((int64_t)b * a.Hq + hq) * ((L + 255) / 256) + xb) * 4 + wave] = nresc_;   // [B][Hq][blocks][4 waves], zeroed by the caller
#ifdef MM355_ATTN4_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tm6 = __builtin_readcyclecounter();
    if (lane == 0 && (wave == 0 || wave == 3) && q0 + 64 <= L) {
        long long* w = (long long*)(lse_base + q0) + (wave == 3 ? 8 : 0);
        w[0] = tm1 - tm0; w[1] = tm2 - tm1; w[2] = tm3 - tm2; w[3] = tm4 - tm3; w[4] = tm5 - tm4; w[5] = tm6 - tm5; w[6] = tw; w[7] = T;
    }
#endif
}

}  // namespace attn4

// variant: 0 = the shipped stream, 1 = the serialised debugging stream (same instructions, every LDS read waited for at once, every MFMA
// followed by 32 wait states); tools/ only
int mm355_attn4_fwd_launch(const attn2::Args& a, int variant, hipStream_t s) {
    const int64_t nblk = (int64_t)((a.L + 255) / 256) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    if (variant == 1) hipLaunchKernelGGL(attn4::fwd_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(attn4::fwd_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return mm_launch_status();
}
