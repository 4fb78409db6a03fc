// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . B[N,K]^T)  -- the nn.Linear form.
//
// Replaces the cuBLAS GEMMs the reference reaches through nn.Linear (see include/mm355.h).
//
// Structure (per workgroup): BM x BN output tile, BK = 64, WM x WN waves, each wave owns a
// (BM/WM) x (BN/WN) sub-tile as 16x16 fragments of v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
//   * A and B tiles live in LDS as [rows][64 bf16] (128-B rows) with the 16-B chunk index XOR-ed with
//     (row & 7): ds_read_b128 fragment reads are bank-conflict free (cdna guide T2).
//   * staging: either register staged (global_load_dwordx4 -> ds_write_b128; handles ragged K) or
//     LDS-DMA (global_load_lds_dwordx4; the swizzle is applied to the per-lane SOURCE address because the
//     LDS destination of the DMA is lane-linear).  Two LDS stages, next tile in flight during the MFMAs.
//   * epilogue: accumulators -> wave-private LDS slab -> row-contiguous 16-B stores with bias / GELU /
//     residual / accumulate fused.
//   * workgroup -> tile map: bijective XCD remap (block b runs on XCD b % 8) then grouped raster so the
//     blocks sharing an XCD's L2 walk neighbouring tiles.
#include "gemm_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

// Shared epilogue: accumulators -> wave-private LDS slab -> row-contiguous 16-B stores with the fused epilogue.
template <int TM, int TN, int FM, int FN>
MM_DEV void gemm_epilogue(f32x4 (&acc)[FM][FN], const GemmArgs& a, unsigned char* smem, int m0, int n0, int wm, int wn, int wave, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, N = a.N;
    constexpr int CPL = TN / 4;                          // columns handled by one lane per row
    float* stg = (float*)smem + wave * (16 * TN);
    const uint32_t fl = a.flags;
    const int row_l = lane >> 2, col_l = (lane & 3) * CPL;
    uint16_t* Cb = (uint16_t*)a.C;
    float* Cf = (float*)a.C;
    const bool vec_ok = ((a.ldc & 7) == 0) && (!(fl & MM355_GEMM_RESIDUAL) || (a.ldr & 7) == 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        // the staging slab is private to this wave and the LDS executes one wave's instructions in order: a wave-level
        // fence (no s_barrier) is all the write -> read -> next write hand-over needs; the caller has already made sure
        // (block barrier) that nobody still reads the tile data this slab overlays
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(fq * 4 + r) * TN + j * 16 + fr] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int grow = m0 + wm * TM + i * 16 + row_l;
        if (grow < M) {
            const int64_t rr = (fl & MM355_GEMM_RESIDUAL) ? (a.res_mod > 0 ? (int64_t)(grow % a.res_mod) : (int64_t)grow) : 0;
#pragma unroll
            for (int j = 0; j < CPL / 8; ++j) {
                const int c = n0 + wn * TN + col_l + j * 8;
                if (c >= N) continue;
                float v[8];
                const f32x4 s0 = *(const f32x4*)(stg + row_l * TN + col_l + j * 8);
                const f32x4 s1 = *(const f32x4*)(stg + row_l * TN + col_l + j * 8 + 4);
                v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w;
                v[4] = s1.x; v[5] = s1.y; v[6] = s1.z; v[7] = s1.w;
                epi_store8(a, fl, vec_ok, grow, rr, c, v);
            }
        }
    }
}

// Epilogue of the fused gate|up GEMM (mm355_gemm_swiglu_bf16): a wave's 64 tile columns are 32 gate channels followed by the SAME 32
// up channels (the B tile is staged from two row ranges of the fused weight, see gemm_pp_tile<.., SWI>), so after the staging slab a
// lane finds gate and up of a channel 32 floats apart in its row: gu goes out as two 64-B runs per wave row (gate block at column ch,
// up block at column I + ch) and act = bf(silu(bf(g))) * bf(u) -- the arithmetic of swiglu_fwd_kernel on the bf16-rounded values, bit
// for bit -- as one 16-B store per lane.  a.C = gu [M][ldc], a.res = act [M][ldr], a.res_mod = I.
MM_DEV void gemm_epilogue_swiglu(f32x4 (&acc)[8][4], const GemmArgs& a, unsigned char* smem, int m0, int tn, int wm, int wn, int wave, int lane) {
    constexpr int TN = 64;
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, I = (int)a.res_mod;
    float* stg = (float*)smem + wave * (16 * TN);
    const int row_l = lane >> 2, k4 = lane & 3;
    uint16_t* gu = (uint16_t*)a.C;
    uint16_t* act = (uint16_t*)a.res;
    const int ch0 = tn * 128 + wn * 32;                       // first channel of this wave
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(fq * 4 + r) * TN + j * 16 + fr] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int grow = m0 + wm * 128 + i * 16 + row_l;
        if (grow < M) {
            // this lane: channels ch0 + k4*8 .. +7 -- gate from slab columns k4*8, up from 32 + k4*8
            float g[8], u[8], o[8];
            const f32x4 g0 = *(const f32x4*)(stg + row_l * TN + k4 * 8), g1 = *(const f32x4*)(stg + row_l * TN + k4 * 8 + 4);
            const f32x4 u0 = *(const f32x4*)(stg + row_l * TN + 32 + k4 * 8), u1 = *(const f32x4*)(stg + row_l * TN + 32 + k4 * 8 + 4);
            g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
            u[0] = u0.x; u[1] = u0.y; u[2] = u0.z; u[3] = u0.w; u[4] = u1.x; u[5] = u1.y; u[6] = u1.z; u[7] = u1.w;
            const u32x4 gp = pack8(g), up = pack8(u);
            const int ch = ch0 + k4 * 8;
            *(u32x4*)(gu + (int64_t)grow * a.ldc + ch) = gp;
            *(u32x4*)(gu + (int64_t)grow * a.ldc + I + ch) = up;
            float gb[8], ub[8];
            unpack8(gp, gb);
            unpack8(up, ub);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = round_bf(gb[e] / (1.0f + __expf(-gb[e]))) * ub[e];
            *(u32x4*)(act + (int64_t)grow * a.ldr + ch) = pack8(o);
        }
    }
}

// Epilogue of the fused down_proj input-gradient GEMM + SwiGLU backward (mm355_gemm_swiglu_bwd_bf16).  The accumulators are
// d act = dY . Wd of a 256 x 256 (rows x channels) tile; rounded to bf16 they are exactly what mm355_gemm_bf16 would have stored and
// mm355_swiglu_bwd_t re-read, and the same arithmetic follows: with g / u from gu [M][2 I] (a.res)
//     a = bf(silu(g)) u,   du = da bf(silu(g)),   dg = da u sigma(g) (1 + g (1 - sigma(g)))
// go out row-major as dgu [M][2 I] (a.C) AND contraction-major as actT [I][ld_aux] (aux0), dguT [2 I][ld_aux] (aux1) -- what the two
// weight-gradient GEMMs read -- through a wave-private LDS tile: 32 rows x 64 channels x 3 quantities per flush, 16-B transposed
// stores (64-B runs per channel and flush, two flushes fill a line).  d act itself never reaches memory.
// LDS: [0, 32 KiB) the waves' fp32 staging slabs, then 8 x SWB_T bytes of transposition tiles (launch with SWB_LDS).
constexpr int SWB_TROW = 144;                                // bytes per tile row: 64 channels bf16 + 16 (16-B aligned rows, banks spread)
constexpr int SWB_T = 3 * 32 * SWB_TROW;                     // 13 824 B per wave
constexpr int SWB_LDS = 32768 + 8 * SWB_T;                   // 143 360 B (of 160 KiB)
MM_DEV void gemm_epilogue_swiglu_bwd(f32x4 (&acc)[8][4], const GemmArgs& a, unsigned char* smem, int m0, int n0, int wm, int wn, int wave, int lane) {
    constexpr int TN = 64, PF = 4;                           // PF: i-blocks of gate / up rows kept in flight ahead of their use
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds4_t;
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, I = a.N;
    float* stg = (float*)smem + wave * (16 * TN);
    unsigned char* tile = smem + 32768 + wave * SWB_T;       // [3 quantities][32 rows][SWB_TROW]
    const int row_l = lane >> 2, k4 = lane & 3;
    const uint16_t* gu = a.res;
    uint16_t* dgu = (uint16_t*)a.C;
    const int c_wave = n0 + wn * 64;                         // first channel of this wave
    if (c_wave >= I) return;                                 // (I % 64 == 0: a wave is in or out as a whole)
    const int row0 = m0 + wm * 128 + row_l;                  // this lane's row in i-block 0
    u32x4 gq[PF][2], uq[PF][2];
    auto fetch = [&](int i, int slot) {
        const int grow = row0 + i * 16;
#pragma unroll
        for (int j8 = 0; j8 < 2; ++j8) {
            const int64_t ro = (int64_t)min(grow, M - 1) * a.ldr + c_wave + k4 * 16 + j8 * 8;
            gq[slot][j8] = *(const u32x4*)(gu + ro);
            uq[slot][j8] = *(const u32x4*)(gu + ro + I);
        }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fetch(i, i);
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int i = ib * 2 + half;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(fq * 4 + r) * TN + j * 16 + fr] = acc[i][j][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int grow = row0 + i * 16;
            const bool live = grow < M;
#pragma unroll
            for (int j8 = 0; j8 < 2; ++j8) {
                const int c = k4 * 16 + j8 * 8;
                float daf[8], g[8], u[8], dg[8], du[8], av[8], da[8];
                const f32x4 s0 = *(const f32x4*)(stg + row_l * TN + c), s1 = *(const f32x4*)(stg + row_l * TN + c + 4);
                daf[0] = s0.x; daf[1] = s0.y; daf[2] = s0.z; daf[3] = s0.w; daf[4] = s1.x; daf[5] = s1.y; daf[6] = s1.z; daf[7] = s1.w;
                unpack8(pack8(daf), da);                     // the bf16 d act the unfused path stores and re-reads
                unpack8(gq[i % PF][j8], g);
                unpack8(uq[i % PF][j8], u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sg = 1.0f / (1.0f + __expf(-g[e]));
                    const float silu = g[e] * sg;
                    av[e] = round_bf(silu) * u[e];
                    du[e] = da[e] * round_bf(silu);
                    dg[e] = da[e] * u[e] * (sg * (1.0f + g[e] * (1.0f - sg)));
                }
                const u32x4 pg = pack8(dg), pu = pack8(du), pa = pack8(av);
                if (live) {
                    const int64_t wo = (int64_t)grow * a.ldc + c_wave + c;
                    *(u32x4*)(dgu + wo) = pg;
                    *(u32x4*)(dgu + wo + I) = pu;
                }
                unsigned char* tp = tile + (half * 16 + row_l) * SWB_TROW + c * 2;
                *(u32x4*)(tp) = pa;
                *(u32x4*)(tp + 32 * SWB_TROW) = pg;
                *(u32x4*)(tp + 64 * SWB_TROW) = pu;
            }
            if (i + PF < 8) fetch(i + PF, i % PF);           // this slot's values are consumed: refill it PF blocks ahead
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // flush 32 rows x 64 channels x 3 quantities with ds_read_b64_tr_b16: the 16 lanes of a group supply sixteen 8-byte pieces
        // P_0 .. P_15 (each four consecutive channels of one tile row) and lane L receives element L % 4 of P_{4k + L / 4}, k = 0 .. 3.
        // With P_{4k + s} = row 8 s + k of a 4-channel block, lane L = 4 s + j ends up with rows 8 s .. 8 s + 3 (second read: + 4 .. 7) of
        // channel j: the four lanes j, 4 + j, 8 + j, 12 + j hold the 32 rows of one channel, i.e. one 64-B run of the transposed output.
        const int r_base = m0 + wm * 128 + ib * 32;
        const int s_l = fr >> 2, j_l = fr & 3;
        const unsigned char* tb = tile + (8 * (fr & 3) + (fr >> 2)) * SWB_TROW + fq * 8;   // this lane's PIECE: row 8 (l % 4) + l / 4
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {                 // channel block it*16 + fq*4
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(tb + q * 32 * SWB_TROW + it * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(tb + (q * 32 + 4) * SWB_TROW + it * 32));
                const int ch = c_wave + it * 16 + fq * 4 + j_l;
                const int r8 = r_base + 8 * s_l;
                if (r8 < M) {                                // (M % 8 == 0: a vector is in or out as a whole)
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    uint16_t* outp = (q == 0 ? a.aux0 : a.aux1) + (int64_t)(q == 2 ? I + ch : ch) * a.ld_aux;
                    *(s16x8*)(outp + r8) = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
        }
    }
}

// Epilogue of the fused QKV projection + RoPE (mm355_gemm_rope_bf16, head size 128).  The B tile is staged from permuted weight rows
// (gemm_pp_tile<.., ROPE>): a wave's 64 tile columns are d = sub*32 .. +31 (first half of a head) followed by d = 64 + sub*32 .. +31 (second
// half), so the rotation partners x1 = x[d], x2 = x[d + 64] of a row sit 32 floats apart in the wave's staging slab.  Both are rounded to
// bf16 first (what mm355_gemm_bf16 would have stored) and rotated with the arithmetic of rope_qk_kernel, bit for bit:
//     y1 = bf(bf(x1 c) + bf(-x2 s)),  y2 = bf(bf(x2 c) + bf(x1 s)),  c / s = cos / sin[position][d]
// for the columns below a.res_mod (the q and k blocks); the v block is stored as it is.  a.aux0 / a.aux1 = cos / sin tables [*][128] bf16,
// a.ld_aux = L (rows per sample), a.bias = optional int32 position offsets per sample.
MM_DEV void gemm_epilogue_rope(f32x4 (&acc)[8][4], const GemmArgs& a, unsigned char* smem, int m0, int n0, int wm, int wn, int wave, int lane) {
    constexpr int TN = 64;
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, L = (int)a.ld_aux, n_rot = (int)a.res_mod;
    float* stg = (float*)smem + wave * (16 * TN);
    const int row_l = lane >> 2, k4 = lane & 3;
    uint16_t* out = (uint16_t*)a.C;
    const int32_t* pos_off = (const int32_t*)a.bias;
    const int d1 = (wn & 1) * 32 + k4 * 8;                    // this lane's eight d of the first half
    const int col1 = n0 + (wn >> 1) * 128 + d1;               // and their output columns (second half: + 64)
    const bool rot = col1 < n_rot;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(fq * 4 + r) * TN + j * 16 + fr] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int grow = m0 + wm * 128 + i * 16 + row_l;
        if (grow < M) {
            float a1[8], a2[8], x1[8], x2[8];
            const f32x4 p0 = *(const f32x4*)(stg + row_l * TN + k4 * 8), p1 = *(const f32x4*)(stg + row_l * TN + k4 * 8 + 4);
            const f32x4 q0 = *(const f32x4*)(stg + row_l * TN + 32 + k4 * 8), q1 = *(const f32x4*)(stg + row_l * TN + 32 + k4 * 8 + 4);
            a1[0] = p0.x; a1[1] = p0.y; a1[2] = p0.z; a1[3] = p0.w; a1[4] = p1.x; a1[5] = p1.y; a1[6] = p1.z; a1[7] = p1.w;
            a2[0] = q0.x; a2[1] = q0.y; a2[2] = q0.z; a2[3] = q0.w; a2[4] = q1.x; a2[5] = q1.y; a2[6] = q1.z; a2[7] = q1.w;
            u32x4 w1 = pack8(a1), w2 = pack8(a2);
            if (rot) {
                unpack8(w1, x1);
                unpack8(w2, x2);
                const int b = grow / L;
                const int l = grow - b * L + (pos_off ? pos_off[b] : 0);
                float c[8], sn[8], y1[8], y2[8];
                unpack8(*(const u32x4*)(a.aux0 + (int64_t)l * 128 + d1), c);
                unpack8(*(const u32x4*)(a.aux1 + (int64_t)l * 128 + d1), sn);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    y1[e] = round_bf(x1[e] * c[e]) + round_bf(-x2[e] * sn[e]);
                    y2[e] = round_bf(x2[e] * c[e]) + round_bf(x1[e] * sn[e]);
                }
                w1 = pack8(y1); w2 = pack8(y2);
            }
            *(u32x4*)(out + (int64_t)grow * a.ldc + col1) = w1;
            *(u32x4*)(out + (int64_t)grow * a.ldc + col1 + 64) = w2;
        }
    }
}

// TNL = true: both operands are stored contraction-major ("TN": A = At[K][M], B = Bt[K][N], C = At^T Bt), which is the
// weight-gradient form dW = dY^T X on the activations as they lie in memory -- no transposed copies.  LDS tiles are then
// [64 k][256] with the MFMA fragments gathered by ds_read_b64_tr_b16 (hardware 4x16 transpose read).
template <int BM, int BN, int WM, int WN, bool GLDS, int PIPE = 0, bool TNL = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(GemmArgs a) {
    constexpr int NT = WM * WN * 64, NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int AV = BM * 8 / NT, BV = BN * 8 / NT;           // 16-B vectors per thread (reg staging)
    constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;           // 1-KiB DMA pieces per wave (glds)
    constexpr int GM = 8;
    static_assert(FM >= 1 && FN >= 2 && (FN % 2) == 0, "tile shape");
    static_assert(AV >= 1 && BV >= 1 && AI >= 1 && BI >= 1, "thread/tile ratio");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- workgroup -> tile --------------------------------------------------------------------
    const int total = a.ntm * a.ntn;
    const int bid = blockIdx.x;
    const int q8 = total >> 3, r8 = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int gsize = GM * a.ntn;
    const int grp = logical / gsize;
    const int first_m = grp * GM;
    const int gm = min(a.ntm - first_m, GM);
    const int in_g = logical - grp * gsize;
    const int tm = first_m + in_g % gm;
    const int tn = in_g / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    // split-K: slice blockIdx.y of K (prompt-pass shapes: a few hundred rows against one weight -- the K loop is a latency chain of one
    // LDS-DMA round trip per 64-wide tile; S slices run as S times as many workgroups side by side); C = fp32 partials [slice][M][ldc]
    const int sl = a.kslice ? (int)blockIdx.y : 0;
    const uint16_t* Ab = a.A + (TNL ? (int64_t)sl * a.kslice * a.lda : (int64_t)sl * a.kslice);
    const uint16_t* Bb = a.B + (TNL ? (int64_t)sl * a.kslice * a.ldb : (int64_t)sl * a.kslice);
    const int M = a.M, N = a.N, K = a.kslice ? min(a.kslice, a.K - sl * a.kslice) : a.K;
    const int nk = (K + 63) >> 6;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int sw0 = ((fq) ^ (fr & 7)) << 4;
    const int sw1 = ((4 + fq) ^ (fr & 7)) << 4;
    const int a_off = (wm * TM + fr) * 128;
    const int b_off = A_BYTES + (wn * TN + fr) * 128;

    u32x4 ra[AV], rb[BV];

    auto gload = [&](int kt) {
        const int k0 = kt << 6;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int v = tid + i * NT, row = v >> 3, c = v & 7;
            const int gr = min(m0 + row, M - 1), k = k0 + c * 8;
            ra[i] = (k < K) ? *(const u32x4*)(Ab + (int64_t)gr * a.lda + k) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < BV; ++i) {
            const int v = tid + i * NT, row = v >> 3, c = v & 7;
            const int gr = min(n0 + row, N - 1), k = k0 + c * 8;
            rb[i] = (k < K) ? *(const u32x4*)(Bb + (int64_t)gr * a.ldb + k) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int v = tid + i * NT, row = v >> 3, c = v & 7;
            *(u32x4*)(sb + row * 128 + ((c ^ (row & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BV; ++i) {
            const int v = tid + i * NT, row = v >> 3, c = v & 7;
            *(u32x4*)(sb + A_BYTES + row * 128 + ((c ^ (row & 7)) << 4)) = rb[i];
        }
    };
    // LDS-DMA sources: one row pointer per 1-KiB piece (8 rows x 128 B), computed once; the wave index is made
    // provably uniform so that the LDS destination (M0) is scalar arithmetic, not a per-DMA v_readfirstlane.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const uint16_t* srcA[AI];
    const uint16_t* srcB[BI];
    if constexpr (!TNL) {
        const int rin = lane >> 3;                           // row inside the 8-row piece
        const int c = (lane & 7) ^ rin;                      // source chunk that belongs in LDS slot (lane & 7)
#pragma unroll
        for (int i = 0; i < AI; ++i)
            srcA[i] = Ab + (int64_t)min(m0 + (i * NW + wave_s) * 8 + rin, M - 1) * a.lda + c * 8;
#pragma unroll
        for (int i = 0; i < BI; ++i)
            srcB[i] = Bb + (int64_t)min(n0 + (i * NW + wave_s) * 8 + rin, N - 1) * a.ldb + c * 8;
    } else {
        // [64 k][256 cols] tiles, 512-B rows: a 1-KiB piece is 2 k-rows; physical 16-B chunk = logical ^ f(row) with
        // f(row) = 2*(row & 3) + 8*((row >> 3) & 1) so that the 8 rows x 2 chunks of one tr-read half-wave hit 16 distinct slots
        static_assert(!TNL || (BM == 256 && BN == 256), "TN layout: 256x256 tile only");
        const int rinT = lane >> 5, pcT = lane & 31;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = (i * NW + wave_s) * 2 + rinT;
            const int c = pcT ^ (2 * (row & 3) + 8 * ((row >> 3) & 1));
            srcA[i] = Ab + (int64_t)row * a.lda + min(m0 + c * 8, M - 8);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int row = (i * NW + wave_s) * 2 + rinT;
            const int c = pcT ^ (2 * (row & 3) + 8 * ((row >> 3) & 1));
            srcB[i] = Bb + (int64_t)row * a.ldb + min(n0 + c * 8, N - 8);
        }
    }
    auto gdma = [&](int kt, int buf) {
        const int k0 = kt << 6;
        const int64_t ka = TNL ? (int64_t)k0 * a.lda : (int64_t)k0, kb = TNL ? (int64_t)k0 * a.ldb : (int64_t)k0;
        unsigned char* sb = smem + buf * STAGE + wave_s * 1024;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + ka), (lptr_t)(sb + i * NW * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + kb), (lptr_t)(sb + A_BYTES + i * NW * 1024), 16, 0, 0);
    };

    // ---- main loop -----------------------------------------------------------------------------
    if constexpr (PIPE == 0) {
    if constexpr (GLDS) {
        gdma(0, 0);
    } else {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            if constexpr (GLDS) gdma(kt + 1, cur ^ 1);
            else gload(kt + 1);
        }
        const unsigned char* sb = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = kk ? sw1 : sw0;
            bf16x8 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(sb + a_off + i * 2048 + sw);
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *(const bf16x8*)(sb + b_off + j * 2048 + sw);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (!GLDS) {
            if (more) lstore(cur ^ 1);
        }
        __syncthreads();
        cur ^= 1;
    }

    } else {
        // Software-pipelined schedule (LDS-DMA staging only): ONE barrier per K tile, placed in the MIDDLE of the
        // tile's MFMAs.  Fragments are double buffered in registers (f0 = k-step 0, f1 = k-step 1), so every
        // ds_read batch is issued in front of a 32-MFMA cluster, and the DMA of tile t+2 is in flight for a whole
        // iteration before anybody waits for it:
        //     read f1(t) | MFMA f0(t) | vmcnt(0)+barrier | DMA tile t+2 -> buffer of t | read f0(t+1) | MFMA f1(t)
        static_assert(GLDS, "pipelined schedule uses LDS-DMA staging");
        static_assert(FM % 2 == 0, "pipelined schedule splits the A fragments in two halves");
        constexpr int HM = FM / 2;
        // four 16-MFMA units per K tile: (k-step, A half) = (0,lo) (0,hi) (1,lo) (1,hi); two rotating A register
        // sets X/Y and two B sets P/Q, each refilled from LDS one unit before it is consumed.
        bf16x8 X[HM], Y[HM], P[FN], Q[FN];
        // TN layout: per-lane byte offsets of the 8-byte transposed-read granules (excluding the k-step immediate)
        int toA[TNL ? FM : 1], toB[TNL ? FN : 1];
        if constexpr (TNL) {
            const int j4 = fr >> 2, q = fr & 3, f = 2 * j4 + 8 * (fq & 1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
                toA[i] = (fq * 8 + j4) * 512 + ((((wm * TM) >> 3) + i * 2 + (q >> 1)) ^ f) * 16 + (q & 1) * 8;
#pragma unroll
            for (int j = 0; j < FN; ++j)
                toB[j] = A_BYTES + (fq * 8 + j4) * 512 + ((((wn * TN) >> 3) + j * 2 + (q >> 1)) ^ f) * 16 + (q & 1) * 8;
        }
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds4_t;
        auto trfrag = [&](const unsigned char* p) -> bf16x8 {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)p);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p + 4 * 512));
            return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto rdA = [&](bf16x8 (&af)[HM], int buf, int half, int kk) {
            const int sw = kk ? sw1 : sw0;
            if constexpr (TNL) {
                const unsigned char* sb = smem + buf * STAGE + kk * 32 * 512;
#pragma unroll
                for (int i = 0; i < HM; ++i) af[i] = trfrag(sb + toA[half * HM + i]);
            } else {
                const unsigned char* sb = smem + buf * STAGE + a_off + half * HM * 2048 + sw;
#pragma unroll
                for (int i = 0; i < HM; ++i) af[i] = *(const bf16x8*)(sb + i * 2048);
            }
        };
        auto rdB = [&](bf16x8 (&bf)[FN], int buf, int kk) {
            const int sw = kk ? sw1 : sw0;
            if constexpr (TNL) {
                const unsigned char* sb = smem + buf * STAGE + kk * 32 * 512;
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[j] = trfrag(sb + toB[j]);
            } else {
                const unsigned char* sb = smem + buf * STAGE + b_off + sw;
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[j] = *(const bf16x8*)(sb + j * 2048);
            }
        };
        auto mm = [&](const bf16x8 (&af)[HM], const bf16x8 (&bf)[FN], int half) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[half * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[half * HM + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        gdma(0, 0);
        __syncthreads();
        if (nk > 1) gdma(1, 1);
        rdA(X, 0, 0, 0);
        rdB(P, 0, 0);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            rdA(Y, cur, 1, 0);                              // unit 0: (k0, lo)
            __builtin_amdgcn_sched_barrier(0);
            mm(X, P, 0);
            __builtin_amdgcn_sched_barrier(0);
            rdA(X, cur, 0, 1);                              // unit 1: (k0, hi)
            rdB(Q, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(Y, P, 1);
            __builtin_amdgcn_sched_barrier(0);
            rdA(Y, cur, 1, 1);                              // unit 2: (k1, lo)
            __builtin_amdgcn_sched_barrier(0);
            mm(X, Q, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                 // tile kt+1 landed everywhere; buffer `cur` fully read
            if (kt + 2 < nk) gdma(kt + 2, cur);
            if (kt + 1 < nk) {                                // unit 3: (k1, hi)
                rdA(X, cur ^ 1, 0, 0);
                rdB(P, cur ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mm(Y, Q, 1);
            __builtin_amdgcn_sched_barrier(0);
            cur ^= 1;
        }
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    if (a.kslice) {                                          // this slice's partial tile, plain fp32 (the host set OUT_F32 and no other flag)
        GemmArgs e = a;
        e.C = (float*)a.C + (int64_t)sl * a.M * a.ldc;
        gemm_epilogue<TM, TN, FM, FN>(acc, e, smem, m0, n0, wm, wn, wave, lane);
        return;
    }
    gemm_epilogue<TM, TN, FM, FN>(acc, a, smem, m0, n0, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// Ring-pipelined 256x256 kernel: the K dimension is consumed in HALF tiles (BK = 32) from a ring of four 32-KiB LDS
// slots.  Measurements (timing-only ablation builds, since removed) suggested that the cost of the staging is the BURST: when all eight waves
// issue their LDS-DMAs right after a barrier they queue in the texture path and no wave issues MFMAs meanwhile.  Here
//   * each wave issues its 4 DMA pieces of half-tile s+3 two at a time IN BETWEEN its 16-MFMA units,
//   * waits are counted (s_waitcnt vmcnt(8): the two most recent half-tiles may still be in flight), never a drain,
//   * a raw s_barrier per half-tile (32 MFMAs per wave) publishes half-tile s+1 and frees slot s for the DMA of s+4.
// LDS half-tile layout: [512 rows (A then B)][32 bf16] (64-B rows); physical 16-B chunk = logical ^ 3*((row>>3)&1), which
// makes every ds_read_b128 fragment read conflict free; the DMA applies the same XOR on its per-lane source address.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(GemmArgs a) {
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16, HM = FM / 2;
    constexpr int SLOT = (BM + BN) * 64;                    // 32 KiB
    constexpr int A_BYTES = BM * 64;
    constexpr int GM = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int total = a.ntm * a.ntn;
    const int bid = blockIdx.x;
    const int q8 = total >> 3, r8 = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int gsize = GM * a.ntn;
    const int grp = logical / gsize;
    const int first_m = grp * GM;
    const int gm = min(a.ntm - first_m, GM);
    const int in_g = logical - grp * gsize;
    const int tm = first_m + in_g % gm;
    const int tn = in_g / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_s / WN, wn = wave_s % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, N = a.N;
    const int nh = a.K >> 5;                                 // half tiles (K % 64 == 0 => nh even, >= 2)

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA sources: half tile = 512 rows x 64 B = 32 pieces of 1 KiB (16 rows each); wave w owns pieces w, w+8, w+16, w+24
    // (pieces 0..15 = A rows, 16..31 = B rows)
    const uint16_t* src[4];
    {
        const int rin = lane >> 2;                           // row inside the 16-row piece
        const int c = (lane & 3) ^ (3 * ((rin >> 3) & 1));   // logical chunk that belongs in physical slot (lane & 3)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = i * NW + wave_s;
            const int row = (piece & 15) * 16 + rin;
            src[i] = (piece < 16) ? a.A + (int64_t)min(m0 + row, M - 1) * a.lda + c * 8
                                  : a.B + (int64_t)min(n0 + row, N - 1) * a.ldb + c * 8;
        }
    }
    auto dma = [&](int s, int i) {                           // piece i of half tile s
        unsigned char* slot = smem + (s & 3) * SLOT + (i * NW + wave_s) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (s << 5)), (lptr_t)slot, 16, 0, 0);
    };
    const int sw = (fq ^ (3 * ((fr >> 3) & 1))) << 4;
    const int a_off = (wm * TM + fr) * 64 + sw;
    const int b_off = A_BYTES + (wn * TN + fr) * 64 + sw;
    bf16x8 X[HM], Y[HM], P[FN];
    auto rdA = [&](bf16x8 (&af)[HM], int s, int half) {
        const unsigned char* sb = smem + (s & 3) * SLOT + a_off + half * HM * 1024;
#pragma unroll
        for (int i = 0; i < HM; ++i) af[i] = *(const bf16x8*)(sb + i * 1024);
    };
    auto rdB = [&](bf16x8 (&bf)[FN], int s) {
        const unsigned char* sb = smem + (s & 3) * SLOT + b_off;
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = *(const bf16x8*)(sb + j * 1024);
    };
    auto mm = [&](const bf16x8 (&af)[HM], const bf16x8 (&bf)[FN], int half) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < HM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[half * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[half * HM + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // one half tile: MODE 0 = steady state (issue s+3, keep 8 in flight), 1 = two tiles still in flight after this one,
    // 2 = last staged tile is the next one (drain), 3 = final half tile (nothing to fetch or publish)
    auto step = [&](auto mode_c, int s, bf16x8 (&Pc)[FN], bf16x8 (&Pn)[FN]) {
        constexpr int MODE = decltype(mode_c)::value;
        rdA(Y, s, 1);
        if constexpr (MODE == 0) { dma(s + 3, 0); dma(s + 3, 1); }
        __builtin_amdgcn_sched_barrier(0);
        mm(X, Pc, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 0) {
            dma(s + 3, 2); dma(s + 3, 3);
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        if constexpr (MODE != 3) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rdA(X, s + 1, 0);
            rdB(Pn, s + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        mm(Y, Pc, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    bf16x8 Q[FN];

    // prologue: half tiles 0, 1, 2 in flight; wait for 0   (host guarantees nh >= 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(2, i);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rdA(X, 0, 0);
    rdB(P, 0);
    int s = 0;
    for (; s < nh - 4; s += 2) {
        step(M0{}, s, P, Q);
        step(M0{}, s + 1, Q, P);
    }
    step(M0{}, s, P, Q);
    step(M1{}, s + 1, Q, P);
    step(M2{}, s + 2, P, Q);
    step(M3{}, s + 3, Q, P);
    __syncthreads();
    gemm_epilogue<TM, TN, FM, FN>(acc, a, smem, m0, n0, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// Ping-pong 256x256x64 kernel.  The eight waves form two groups (wave w and w + 4 share a SIMD; group = M half of the tile)
// that run ONE BARRIER APART: while one group issues its 16-MFMA cluster (one 64x32 quadrant of its 128x64 output x K = 64),
// the other does everything else (fragment ds_reads for its next cluster, two LDS-DMA pieces, counted waits), then they swap.
// The matrix pipe of every SIMD always has exactly one wave feeding it and nothing else competes for issue in that wave.
//   phase p of a K tile (per wave):  quadrant   fragments read in the load section      LDS-DMA issued (one 16-KiB quarter tile)
//        0                            (a0, b0)   B cols block 0 of THIS tile             quarter phi + 6 in need order
//        1                            (a0, b1)   B cols block 1
//        2                            (a1, b1)   A rows block 1
//        3                            (a1, b0)   A rows block 0 of the NEXT tile
// A K tile is staged as four quarters in the order they are first needed (A rows 0-63 of both halves, B cols 0-31 of each wave
// column, B cols 32-63, A rows 64-127) into a two-tile ring; quarter s is issued in phase s - 6, needed in phase s - 1, and
// overwrites quarter s - 8, whose last read was >= 3 phases earlier.  Waits are counted (vmcnt(8): four quarters stay in
// flight), placed one phase before the first read of the retired quarter with a barrier in between (both groups).
// LDS image per tile as in the other LDS-DMA variants: [256 rows][64 k] for A then B, 128-B rows, 16-B chunk ^ (row & 7).
// ------------------------------------------------------------------------------------------------
template <int N> MM_DEV void wait_vmcnt() {
#if defined(MM355_SWB_ABL) && MM355_SWB_ABL == 3
    static_assert(N >= 0 && N <= 10, "unsupported count");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#else
    static_assert(N == 0 || N == 2 || N == 4 || N == 6 || N == 8, "unsupported count");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// body of one 256x256 output tile; `bid` = index of the workgroup within ITS problem (the pair kernel below runs two problems
// in one grid)
template <bool TA, bool TB, int ABL = 0, bool SWI = false, bool SWB = false, bool ROPE = false>
MM_DEV void gemm_pp_tile(const GemmArgs& a, const int bid, unsigned char* smem) {
    static_assert(!(SWI || SWB || ROPE) || (!TA && !TB), "fused epilogues: row-major operands");
    constexpr int BM = 256, BN = 256, TM = 128, TN = 64, FM = 8, FN = 4;
    constexpr int BUF = (BM + BN) * 128;                    // 64 KiB per K tile
    constexpr int A_BYTES = BM * 128;

    const int total = a.ntm * a.ntn;
    const int q8 = total >> 3, r8 = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int GM = a.gm;
    const int gsize = GM * a.ntn;
    const int grp = logical / gsize;
    const int first_m = grp * GM;
    const int gm = min(a.ntm - first_m, GM);
    const int in_g = logical - grp * gsize;
    const int tm = first_m + in_g % gm;
    const int tn = in_g / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_s >> 2, wn = wave_s & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, N = a.N;
    const int nk = a.K >> 6;                                 // host guarantees K % 128 == 0, K >= 128
#ifdef MM355_SWB_STAGGER                                     // experiment: the first wave of workgroups starts in MM355_SWB_PHASES phases
    if constexpr (SWB) {
        if (bid < 256) {
            const int ph = (bid >> 3) % MM355_SWB_PHASES;
            for (int i = 0; i < ph * MM355_SWB_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
#endif

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- LDS-DMA: quarter kinds in need order 0: A rows block 0, 1: B cols block 0, 2: B cols block 1, 3: A rows block 1;
    //      16 pieces of 8 rows each, this wave moves pieces wave and wave + 8 of every quarter
    uint32_t src[4][2];                                      // per-lane source byte offset from the buffer base
    int dst[4][2];                                           // tile-relative LDS byte offset of the piece (wave-uniform)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = wave_s + 8 * h;
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) {
            const bool isA = kd == 0 || kd == 3;
            const int blk = isA ? (kd == 3) : (kd - 1);      // rows / cols block 0 or 1 of the wave tile
            if (isA && !TA) {                                // [256 m][64 k] image, piece = 8 rows x 128 B
                const int rin = lane >> 3, c = (lane & 7) ^ rin;
                const int row = h * 128 + blk * 64 + wave_s * 8;
                src[kd][h] = (uint32_t)((int64_t)(min(m0 + row + rin, M - 1) - m0) * a.lda * 2 + c * 16);   // from row m0
                dst[kd][h] = row * 128;
            } else if (isA && TA) {                          // four [64 k][64 m] images (128-B rows), piece = 8 k rows
                const int krow = wave_s * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (2 * ((krow >> 1) & 1) + 4 * ((krow >> 3) & 1));
                src[kd][h] = (uint32_t)((int64_t)krow * a.lda * 2 + min(m0 + h * 128 + blk * 64 + c * 8, M - 8) * 2);
                dst[kd][h] = (h * 2 + blk) * 8192 + wave_s * 1024;
            } else if (!TB) {                                // [256 n][64 k] image
                const int rin = lane >> 3, c = (lane & 7) ^ rin;
                const int row = (q >> 2) * 64 + blk * 32 + (q & 3) * 8;
                if constexpr (SWI) {
                    // tile column block (wave column q >> 2, half blk) = channels tn*128 + (q >> 2)*32 + .. of the gate (blk 0) or
                    // the up (blk 1) rows of the fused [2 I][K] weight; offsets from row 0 of the weight (host: 2 I ldb 2 < 2 GiB)
                    const int wrow = blk * (int)a.res_mod + tn * 128 + (q >> 2) * 32 + (q & 3) * 8 + rin;
                    src[kd][h] = (uint32_t)((int64_t)wrow * a.ldb * 2 + c * 16);
                } else if constexpr (ROPE) {
                    // wave column q >> 2 = (head of the tile, half-of-half sub): its block 0 = weight rows d = sub*32 .. of that head,
                    // block 1 = rows d + 64 (the rotation partners); N % 256 == 0, offsets from the tile's first row n0
                    const int wrow = ((q >> 2) >> 1) * 128 + blk * 64 + ((q >> 2) & 1) * 32 + (q & 3) * 8 + rin;
                    src[kd][h] = (uint32_t)((int64_t)wrow * a.ldb * 2 + c * 16);
                } else {
                    src[kd][h] = (uint32_t)((int64_t)(min(n0 + row + rin, N - 1) - n0) * a.ldb * 2 + c * 16);   // from row n0
                }
                dst[kd][h] = A_BYTES + row * 128;
            } else {                                         // eight [64 k][32 n] images (64-B rows), piece = 16 k rows
                const int krow = (q & 3) * 16 + (lane >> 2);
                const int c = (lane & 3) ^ (2 * ((krow >> 3) & 1));
                src[kd][h] = (uint32_t)((int64_t)krow * a.ldb * 2 + min(n0 + (q >> 2) * 64 + blk * 32 + c * 8, N - 8) * 2);
                dst[kd][h] = A_BYTES + ((q >> 2) * 2 + blk) * 4096 + (q & 3) * 1024;
            }
        }
    }
    // buffer-addressed DMA (resource in SGPRs + 32-bit lane offset + scalar K offset): no 64-bit per-lane address arithmetic.
    // A row-major operand is based at the tile's first row (one scalar 64-bit multiply-add per workgroup), so the 31-bit
    // offsets only span 256 rows + the K extent and the operand itself may be of any size; contraction-major operands are
    // based at the matrix (their offsets run over K rows: pp_eligible keeps those below 2 GiB).
    const uint16_t* baseA = TA ? a.A : a.A + (int64_t)m0 * a.lda;
    const uint16_t* baseB = (TB || SWI) ? a.B : a.B + (int64_t)n0 * a.ldb;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, 0x7fffffff, 0x00020000);
    const int kstepA = TA ? (int)(a.lda * 128) : 128, kstepB = TB ? (int)(a.ldb * 128) : 128;   // bytes per K tile
    auto issue = [&](int kd, int tile) {                     // quarter kd of K tile `tile`
        unsigned char* sb = smem + (tile & 1) * BUF;
        const bool isA = kd == 0 || kd == 3;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsB, (lptr_t)(sb + dst[kd][h]), 16, src[kd][h],
                                                     (ABL == 3 ? (tile & 1) : tile) * (isA ? kstepA : kstepB), 0, 0);   // ABL 3: K tiles 0 / 1 over and over
    };

    // ---- fragments: row-major operands by ds_read_b128, contraction-major ones by two ds_read_b64_tr_b16 (k rows j..j+3 and
    //      j+4..j+7 of a [k][16 cols] block; chunk swizzles f(k) chosen so that a 32-lane gather touches every bank once)
    const int sw0 = ((fq) ^ (fr & 7)) << 4;
    const int sw1 = ((4 + fq) ^ (fr & 7)) << 4;
    const int a_off = (wm * TM + fr) * 128;
    const int b_off = A_BYTES + (wn * TN + fr) * 128;
    int toA[4], toB[2];
    {
        const int j4 = fr >> 2, q = fr & 3;
        const int fa = 2 * ((j4 >> 1) & 1) + 4 * (fq & 1), fb = 2 * (fq & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) toA[i] = (fq * 8 + j4) * 128 + (((i * 2 + (q >> 1)) ^ fa) << 4) + (q & 1) * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) toB[j] = (fq * 8 + j4) * 64 + (((j * 2 + (q >> 1)) ^ fb) << 4) + (q & 1) * 8;
    }
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds4_t;
    auto trfrag = [&](const unsigned char* p, int hi_off) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p + hi_off));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    bf16x8 A[2][4][2], B[2][2][2];
    auto rdA = [&](int ah, int tile) {
        if constexpr (!TA) {
            const unsigned char* sb = smem + (tile & 1) * BUF + a_off + ah * 8192;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                A[ah][i][0] = *(const bf16x8*)(sb + i * 2048 + sw0);
                A[ah][i][1] = *(const bf16x8*)(sb + i * 2048 + sw1);
            }
        } else {
            const unsigned char* sb = smem + (tile & 1) * BUF + (wm * 2 + ah) * 8192;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                A[ah][i][0] = trfrag(sb + toA[i], 4 * 128);
                A[ah][i][1] = trfrag(sb + toA[i] + 32 * 128, 4 * 128);
            }
        }
    };
    auto rdB = [&](int bh, int tile) {
        if constexpr (!TB) {
            const unsigned char* sb = smem + (tile & 1) * BUF + b_off + bh * 4096;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                B[bh][j][0] = *(const bf16x8*)(sb + j * 2048 + sw0);
                B[bh][j][1] = *(const bf16x8*)(sb + j * 2048 + sw1);
            }
        } else {
            const unsigned char* sb = smem + (tile & 1) * BUF + A_BYTES + (wn * 2 + bh) * 4096;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                B[bh][j][0] = trfrag(sb + toB[j], 4 * 64);
                B[bh][j][1] = trfrag(sb + toB[j] + 32 * 64, 4 * 64);
            }
        }
    };
#if defined(MM355_SWB_ABL) && MM355_SWB_ABL == 3
    // TIMING-ONLY (tools/build_swb_abl.sh 3): the epilogue's HBM traffic of ONE tile -- 32 16-B loads of gate / up rows, 80 16-B stores of
    // dgu / act^T / dgu^T per thread, at the epilogue's own addresses -- issued INSIDE the K loop, spread evenly over its 256 phases: what a
    // perfectly overlapped (persistent) kernel would put on the memory system while its matrix pipe runs.  Results are garbage by construction.
    u32x4 swb_sink = u32x4{0u, 0u, 0u, 0u};
    auto swb_extra = [&](int pidx) -> int {                  // returns the number of vector-memory ops it issued (0, 1 or 2)
        int n = 0;
        const int row_l = lane >> 2, k4 = lane & 3;
        const int c_wave = n0 + wn * 64;
        if ((pidx & 7) == 0) {                               // load li = pidx / 8 of 32: (i-block, 8-channel half, gate | up)
            const int li = pidx >> 3, i = li >> 2, j8 = (li >> 1) & 1, gsel = li & 1;
            const int grow = min(m0 + wm * 128 + row_l + i * 16, M - 1);
            const uint16_t* pa = a.res + (int64_t)grow * a.ldr + min(c_wave, a.N - 64) + k4 * 16 + j8 * 8 + gsel * a.N;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(swb_sink) : "v"(pa) : "memory");
            ++n;
        }
        const int s_now = (pidx * 5) >> 4, s_prev = pidx ? ((pidx - 1) * 5) >> 4 : -1;
        if (s_now != s_prev) {                               // store si of 80: 32 row-major dgu vectors, then 48 transposed 16-B runs
            const int si = s_now;
            uint16_t* pa;
            if (si < 32) {
                const int i = si >> 2, j8 = (si >> 1) & 1, gsel = si & 1;
                const int grow = min(m0 + wm * 128 + row_l + i * 16, M - 1);
                pa = (uint16_t*)a.C + (int64_t)grow * a.ldc + min(c_wave, a.N - 64) + k4 * 16 + j8 * 8 + gsel * a.N;
            } else {
                const int t = si - 32, q = t >> 4, it = (t >> 2) & 3, ib = t & 3;
                const int ch = min(c_wave, a.N - 64) + it * 16 + fq * 4 + (fr & 3);
                const int r8 = min(m0 + wm * 128 + ib * 32 + 8 * (fr >> 2), M - 8);
                pa = (q == 0 ? a.aux0 : a.aux1) + (int64_t)(q == 2 ? a.N + ch : ch) * a.ld_aux + r8;
            }
            asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(pa), "v"(swb_sink) : "memory");
            ++n;
        }
        return n;
    };
#endif
    // one phase; P = phase of the K tile, VM = vmcnt to keep in flight, ISSUE: stage quarter (P + 2) & 3 of tile st,
    // NEXTA: tile + 1 exists (read its A rows block 0 in phase 3)
    auto phase = [&](auto p_c, auto vm_c, auto issue_c, auto nexta_c, int tile, int st) {
        constexpr int P = decltype(p_c)::value, VM = decltype(vm_c)::value;
        constexpr bool ISSUE = decltype(issue_c)::value, NEXTA = decltype(nexta_c)::value;
        constexpr int ah = P >> 1, bh = (P == 1 || P == 2) ? 1 : 0;
        if constexpr (ABL != 2) {
            if constexpr (P == 0) rdB(0, tile);
            if constexpr (P == 1) rdB(1, tile);
            if constexpr (P == 2) rdA(1, tile);
            if constexpr (P == 3 && NEXTA) rdA(0, tile + 1);
        }
        if constexpr (ISSUE && ABL != 1) issue((P + 2) & 3, st);
#if defined(MM355_SWB_ABL) && MM355_SWB_ABL == 3
        if constexpr (SWB) {
            // the extra ops sit BEHIND this phase's DMA pieces in the in-order counter: they may stay in flight (VM + n), the pieces older
            // than them are waited for exactly as before
            const int nx = swb_extra(tile * 4 + P);
            if (nx == 0) wait_vmcnt<VM>();
            else if (nx == 1) wait_vmcnt<VM + 1>();
            else wait_vmcnt<VM + 2>();
        } else
#endif
        if constexpr (ABL != 1) wait_vmcnt<VM>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ah * 4 + i][bh * 2 + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ah][i][kk], B[bh][j][kk], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I6 = std::integral_constant<int, 6>;
    using I8 = std::integral_constant<int, 8>;
    using T = std::true_type; using F = std::false_type;

#if defined(MM355_SWB_ABL) && MM355_SWB_ABL == 2
    if constexpr (SWB) {                                     // TIMING-ONLY: no K loop at all -- the epilogue's 896 KB per tile alone
        __syncthreads();
        gemm_epilogue_swiglu_bwd(acc, a, smem, m0, n0, wm, wn, wave, lane);
        return;
    }
#endif
    // prologue: quarters 0..5 (tile 0 complete, tile 1 first half) in flight; quarters 0, 1 landed; A rows block 0 of tile 0
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) issue(kd, 0);
    issue(0, 1);
    issue(1, 1);
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    rdA(0, 0);
    if (wm == 1) __builtin_amdgcn_s_barrier();               // the second group runs one barrier behind the first

    int t = 0;
    for (; t + 2 < nk; t += 2) {                             // steady state: two K tiles = eight phases per trip
        phase(I0{}, I8{}, T{}, T{}, t, t + 1);
        phase(I1{}, I8{}, T{}, T{}, t, t + 1);
        phase(I2{}, I8{}, T{}, T{}, t, t + 2);
        phase(I3{}, I8{}, T{}, T{}, t, t + 2);
        phase(I0{}, I8{}, T{}, T{}, t + 1, t + 2);
        phase(I1{}, I8{}, T{}, T{}, t + 1, t + 2);
        phase(I2{}, I8{}, T{}, T{}, t + 1, t + 3);
        phase(I3{}, I8{}, T{}, T{}, t + 1, t + 3);
    }
    // last two K tiles: the last two quarters go out, then the ring drains
    phase(I0{}, I8{}, T{}, T{}, t, t + 1);
    phase(I1{}, I8{}, T{}, T{}, t, t + 1);
    phase(I2{}, I6{}, F{}, T{}, t, 0);
    phase(I3{}, I4{}, F{}, T{}, t, 0);
    phase(I0{}, I2{}, F{}, T{}, t + 1, 0);
    phase(I1{}, I0{}, F{}, T{}, t + 1, 0);
    phase(I2{}, I0{}, F{}, T{}, t + 1, 0);
    phase(I3{}, I0{}, F{}, F{}, t + 1, 0);
    if (wm == 0) __builtin_amdgcn_s_barrier();               // the first group catches the barrier count up
    __syncthreads();
#if defined(MM355_SWB_ABL) && (MM355_SWB_ABL == 1 || MM355_SWB_ABL == 3)
    if constexpr (SWB) {                                     // TIMING-ONLY: the K loop alone (1) / with the epilogue's traffic inside it (3); the
        float keep = 0.f;                                    // accumulators are consumed so that the MFMAs stay
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 1.2345e-30f) ((float*)a.C)[tid] = keep;
        return;
    }
#endif
    if constexpr (SWI) gemm_epilogue_swiglu(acc, a, smem, m0, tn, wm, wn, wave, lane);
    else if constexpr (SWB) gemm_epilogue_swiglu_bwd(acc, a, smem, m0, n0, wm, wn, wave, lane);
    else if constexpr (ROPE) gemm_epilogue_rope(acc, a, smem, m0, n0, wm, wn, wave, lane);
    else gemm_epilogue<TM, TN, FM, FN>(acc, a, smem, m0, n0, wm, wn, wave, lane);
}

__global__ __launch_bounds__(512) void gemm_pp_swiglu_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemm_pp_tile<false, false, 0, true>(a, blockIdx.x, smem);
}

__global__ __launch_bounds__(512) void gemm_pp_rope_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemm_pp_tile<false, false, 0, false, false, true>(a, blockIdx.x, smem);
}

__global__ __launch_bounds__(512) void gemm_pp_swiglu_bwd_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemm_pp_tile<false, false, 0, false, true>(a, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------------
// Two-phase form of the ping-pong tile (row-major operands): the same two wave groups one barrier apart, but a K tile is TWO phases of
// 32 MFMAs instead of four of 16, i.e. half as many hand-overs between the groups per unit of matrix work:
//        phase   quadrants                 fragments read in the load section          LDS-DMA issued in the load section
//        X(t)    (a0,b0) (a0,b1)           B cols blocks 0, 1 of tile t (8 reads)      A rows block 0 of t+2, A rows block 1 of t+1
//        Y(t)    (a1,b1) (a1,b0)           A rows block 1 of t, block 0 of t+1 (16)    B cols blocks 0, 1 of t+2
// Ring = two K tiles as in gemm_pp_tile; a quarter is re-issued in the first load section after its last reader (of either group)
// has passed, is waited for (counted vmcnt(8): the two quarters of this phase and of the one before stay in flight) at the end of
// the load section that precedes its first read, and is published by the barrier that follows.  Same accumulation order per output
// element as gemm_pp_tile: bit-identical results.
// ------------------------------------------------------------------------------------------------
MM_DEV void gemm_pp2_tile(const GemmArgs& a, const int bid, unsigned char* smem) {
    constexpr int BM = 256, BN = 256, TM = 128, TN = 64, FM = 8, FN = 4;
    constexpr int BUF = (BM + BN) * 128;
    constexpr int A_BYTES = BM * 128;

    const int total = a.ntm * a.ntn;
    const int q8 = total >> 3, r8 = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int GM = a.gm;
    const int gsize = GM * a.ntn;
    const int grp = logical / gsize;
    const int first_m = grp * GM;
    const int gm = min(a.ntm - first_m, GM);
    const int in_g = logical - grp * gsize;
    const int tm = first_m + in_g % gm;
    const int tn = in_g / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_s >> 2, wn = wave_s & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, N = a.N;
    const int nk = a.K >> 6;                                 // host guarantees K % 128 == 0, K >= 128

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // quarter kinds as in gemm_pp_tile: 0 = A rows block 0, 1 = B cols block 0, 2 = B cols block 1, 3 = A rows block 1
    uint32_t src[4][2];
    int dst[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = wave_s + 8 * h;
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) {
            const bool isA = kd == 0 || kd == 3;
            const int blk = isA ? (kd == 3) : (kd - 1);
            const int rin = lane >> 3, c = (lane & 7) ^ rin;
            if (isA) {
                const int row = h * 128 + blk * 64 + wave_s * 8;
                src[kd][h] = (uint32_t)((int64_t)(min(m0 + row + rin, M - 1) - m0) * a.lda * 2 + c * 16);
                dst[kd][h] = row * 128;
            } else {
                const int row = (q >> 2) * 64 + blk * 32 + (q & 3) * 8;
                src[kd][h] = (uint32_t)((int64_t)(min(n0 + row + rin, N - 1) - n0) * a.ldb * 2 + c * 16);
                dst[kd][h] = A_BYTES + row * 128;
            }
        }
    }
    const uint16_t* baseA = a.A + (int64_t)m0 * a.lda;
    const uint16_t* baseB = a.B + (int64_t)n0 * a.ldb;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, 0x7fffffff, 0x00020000);
    auto issue = [&](int kd, int tile) {
        unsigned char* sb = smem + (tile & 1) * BUF;
        const bool isA = kd == 0 || kd == 3;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsB, (lptr_t)(sb + dst[kd][h]), 16, src[kd][h], tile * 128, 0, 0);
    };

    const int sw0 = ((fq) ^ (fr & 7)) << 4;
    const int sw1 = ((4 + fq) ^ (fr & 7)) << 4;
    const int a_off = (wm * TM + fr) * 128;
    const int b_off = A_BYTES + (wn * TN + fr) * 128;
    bf16x8 A[2][4][2], B[2][2][2];
    auto rdA = [&](int ah, int tile) {
        const unsigned char* sb = smem + (tile & 1) * BUF + a_off + ah * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A[ah][i][0] = *(const bf16x8*)(sb + i * 2048 + sw0);
            A[ah][i][1] = *(const bf16x8*)(sb + i * 2048 + sw1);
        }
    };
    auto rdB = [&](int bh, int tile) {
        const unsigned char* sb = smem + (tile & 1) * BUF + b_off + bh * 4096;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            B[bh][j][0] = *(const bf16x8*)(sb + j * 2048 + sw0);
            B[bh][j][1] = *(const bf16x8*)(sb + j * 2048 + sw1);
        }
    };
    // one phase.  YP: false = X, true = Y;  VM = vmcnt kept in flight;  I0 / I1: issue the phase's first / second quarter
    // (X: A0 of t+2 / A1 of t+1;  Y: B0 / B1 of t+2);  NEXTA: tile t+1 exists (Y reads its A rows block 0)
    auto phase = [&](auto y_c, auto vm_c, auto i0_c, auto i1_c, auto nexta_c, int t) {
        constexpr bool YP = decltype(y_c)::value, I0 = decltype(i0_c)::value, I1 = decltype(i1_c)::value, NEXTA = decltype(nexta_c)::value;
        constexpr int VM = decltype(vm_c)::value;
        constexpr int ah = YP ? 1 : 0;
        if constexpr (!YP) {
            rdB(0, t);
            rdB(1, t);
            if constexpr (I0) issue(0, t + 2);
            if constexpr (I1) issue(3, t + 1);
        } else {
            rdA(1, t);
            if constexpr (NEXTA) rdA(0, t + 1);
            if constexpr (I0) issue(1, t + 2);
            if constexpr (I1) issue(2, t + 2);
        }
        wait_vmcnt<VM>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int bh = YP ? 1 - hb : hb;                 // X: (a0,b0) (a0,b1);  Y: (a1,b1) (a1,b0)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][bh * 2 + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ah][i][kk], B[bh][j][kk], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    using I0_ = std::integral_constant<int, 0>; using I2_ = std::integral_constant<int, 2>;
    using I6_ = std::integral_constant<int, 6>; using I8_ = std::integral_constant<int, 8>;
    using T = std::true_type; using F = std::false_type;

    // prologue: tiles 0 and 1 in flight except A rows block 1 of tile 1 (issued by X(0)); first needed: A0, B0, B1 of tile 0
    issue(0, 0); issue(1, 0); issue(2, 0);
    issue(3, 0); issue(0, 1);
    issue(1, 1); issue(2, 1);
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    rdA(0, 0);
    // X(0) re-issues the A rows block 0 slot of buffer 0 (for tile 2) right away: every wave's read of it must have completed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();               // the second group runs one barrier behind the first

    int t = 0;
    for (; t + 2 < nk; ++t) {                                // steady state
        phase(F{}, I8_{}, T{}, T{}, T{}, t);
        phase(T{}, I8_{}, T{}, T{}, T{}, t);
    }
    // t = nk - 2: only A rows block 1 of the last tile is still to be issued; then the ring drains
    phase(F{}, I6_{}, F{}, T{}, T{}, t);
    phase(T{}, I2_{}, F{}, F{}, T{}, t);
    phase(F{}, I0_{}, F{}, F{}, F{}, t + 1);
    phase(T{}, I0_{}, F{}, F{}, F{}, t + 1);
    if (wm == 0) __builtin_amdgcn_s_barrier();               // the first group catches the barrier count up
    __syncthreads();
    gemm_epilogue<TM, TN, FM, FN>(acc, a, smem, m0, n0, wm, wn, wave, lane);
}

__global__ __launch_bounds__(512) void gemm_pp2_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemm_pp2_tile(a, blockIdx.x, smem);
}

template <bool TA, bool TB, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef MM355_PP_PERSIST                                      // TIMING experiment (tools/build_pp_persist.sh): one workgroup per CU walks the tiles
    const int total = a.ntm * a.ntn;
    for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
        gemm_pp_tile<TA, TB, ABL>(a, bid, smem);
        __syncthreads();
    }
#else
    gemm_pp_tile<TA, TB, ABL>(a, blockIdx.x, smem);
#endif
}

// Two independent row-major problems in ONE grid: workgroups [0, n0) work on a0, the rest on a1.  A launch is executed in
// waves of 256 workgroups (one 256x256 tile per CU), so two launches of 384 and 896 tiles cost 2 + 4 wave times while the pair
// costs 5 (LLaMA-3-8B weight gradients of qkv and down_proj).  Each workgroup runs exactly one of the two inlined bodies.
__global__ __launch_bounds__(512) void gemm_pp_pair_kernel(GemmArgs a0, GemmArgs a1, int n0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bid = blockIdx.x;
    if (bid < n0) gemm_pp_tile<false, false>(a0, bid, smem);
    else gemm_pp_tile<false, false>(a1, bid - n0, smem);
}

template <int BM, int BN, int WM, int WN, bool GLDS, int PIPE, bool TNL>
int launch_gemm(GemmArgs a, hipStream_t s);

// TA / TB: operand given contraction-major ([K][M] / [K][N]); eligibility of the K extent and the 31-bit byte offsets is the
// caller's business (pp_eligible)
constexpr int PP_LDS = 2 * (256 + 256) * 128;               // 128 KiB

// tile grid + raster knobs of one ping-pong problem; returns its tile count (<= 0: not launchable)
int64_t pp_prepare(GemmArgs& a) {
    a.ntm = (a.M + 255) / 256;
    a.ntn = (a.N + 255) / 256;
    a.gm = 4;                                               // 4 x ntn raster groups: sweep on LLaMA-3-8B shapes (2/4/8/16/32; profiles/r1_*)
    return (int64_t)a.ntm * a.ntn;
}

template <bool TA, bool TB, int ABL = 0>
int launch_gemm_pp_t(GemmArgs a, hipStream_t s) {
    static std::atomic<uint64_t> lds_ok{0};              // per-device opt-in to > 64 KiB of dynamic LDS
    if (mm_ensure_dynamic_lds((const void*)gemm_pp_kernel<TA, TB, ABL>, PP_LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    const int64_t total = pp_prepare(a);
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
#ifdef MM355_PP_PERSIST
    hipLaunchKernelGGL((gemm_pp_kernel<TA, TB, ABL>), dim3((unsigned)std::min<int64_t>(total, MM355_PP_PERSIST)), dim3(512), PP_LDS, s, a);
#else
    hipLaunchKernelGGL((gemm_pp_kernel<TA, TB, ABL>), dim3((unsigned)total), dim3(512), PP_LDS, s, a);
#endif
    return mm_launch_status();
}

int launch_gemm_pp_pair(GemmArgs a0, GemmArgs a1, hipStream_t s) {
    static std::atomic<uint64_t> lds_ok{0};              // per-device opt-in to > 64 KiB of dynamic LDS
    if (mm_ensure_dynamic_lds((const void*)gemm_pp_pair_kernel, PP_LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    const int64_t n0 = pp_prepare(a0), n1 = pp_prepare(a1);
    if (n0 <= 0 || n1 <= 0 || n0 + n1 > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(gemm_pp_pair_kernel, dim3((unsigned)(n0 + n1)), dim3(512), PP_LDS, s, a0, a1, (int)n0);
    return mm_launch_status();
}

// whole pairs of K tiles, and every source byte offset below 2 GiB: (K rows * ld + M) for a contraction-major operand, based
// at the matrix; (256 rows * ld + K) for a row-major one, based at the tile's first row
bool pp_eligible(const GemmArgs& a, bool ta, bool tb) {
    if (a.K < 128 || (a.K & 127)) return false;
    const int64_t ea = ta ? ((int64_t)a.K * a.lda + a.M) * 2 : (256 * a.lda + a.K) * 2;
    const int64_t eb = tb ? ((int64_t)a.K * a.ldb + a.N) * 2 : (256 * a.ldb + a.K) * 2;
    if (ta && (a.M < 8 || (a.M & 7))) return false;
    if (tb && (a.N < 8 || (a.N & 7))) return false;
    return ea < 0x7fffffffLL && eb < 0x7fffffffLL;
}

#ifdef MM355_LEGACY_VARIANTS
int launch_gemm_pp2(GemmArgs a, hipStream_t s) {
    if (!pp_eligible(a, false, false)) return launch_gemm<256, 256, 2, 4, true, 1, false>(a, s);
    static std::atomic<uint64_t> lds_ok{0};
    if (mm_ensure_dynamic_lds((const void*)gemm_pp2_kernel, PP_LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    const int64_t total = pp_prepare(a);
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(gemm_pp2_kernel, dim3((unsigned)total), dim3(512), PP_LDS, s, a);
    return mm_launch_status();
}

#endif

int launch_gemm_pp(GemmArgs a, hipStream_t s) {
    if (!pp_eligible(a, false, false)) return launch_gemm<256, 256, 2, 4, true, 1, false>(a, s);
    return launch_gemm_pp_t<false, false>(a, s);
}

#ifdef MM355_LEGACY_VARIANTS
int launch_gemm_ring(GemmArgs a, hipStream_t s) {
    if (a.K < 128) return launch_gemm<256, 256, 2, 4, true, 0, false>(a, s);
    constexpr int LDS = 4 * (256 + 256) * 64;               // 128 KiB
    static std::atomic<uint64_t> lds_ok{0};              // per-device opt-in to > 64 KiB of dynamic LDS
    if (mm_ensure_dynamic_lds((const void*)gemm_nt_ring_kernel, LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    a.ntm = (a.M + 255) / 256;
    a.ntn = (a.N + 255) / 256;
    const int64_t total = (int64_t)a.ntm * a.ntn;
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(gemm_nt_ring_kernel, dim3((unsigned)total), dim3(512), LDS, s, a);
    return mm_launch_status();
}

#endif

template <int BM, int BN, int WM, int WN, bool GLDS, int PIPE = 0, bool TNL = false>
int launch_gemm(GemmArgs a, hipStream_t s) {
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int LDS = 2 * STAGE;
    auto kern = gemm_nt_kernel<BM, BN, WM, WN, GLDS, PIPE, TNL>;
    static std::atomic<uint64_t> lds_ok{0};              // per-device opt-in to > 64 KiB of dynamic LDS
    if (mm_ensure_dynamic_lds((const void*)kern, LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    a.ntm = (a.M + BM - 1) / BM;
    a.ntn = (a.N + BN - 1) / BN;
    const int64_t total = (int64_t)a.ntm * a.ntn;
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(WM * WN * 64), LDS, s, a);
    return mm_launch_status();
}

// ------------------------------------------------------------------------------------------------
// transpose: out[c][r] = in[r][c], 64x64 tiles through LDS, 16-B global accesses on both sides.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ in, int64_t ld_in, int rows, int cols,
                                                        uint16_t* __restrict__ out, int64_t ld_out) {
    __shared__ uint16_t tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    // load: 64 rows x 8 vectors
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256, r = v >> 3, c = (v & 7) * 8;
        const int gr = r0 + r, gc = c0 + c;
        uint16_t tmp[8];
        if (gr < rows && gc + 8 <= cols) {
            *(u32x4*)tmp = *(const u32x4*)(in + (int64_t)gr * ld_in + gc);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = (gr < rows && gc + e < cols) ? in[(int64_t)gr * ld_in + gc + e] : (uint16_t)0;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][c + e] = tmp[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256, c = v >> 3, r = (v & 7) * 8;   // output row = input col c
        const int gc = c0 + c, gr = r0 + r;
        if (gc >= cols) continue;
        uint16_t tmp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tmp[e] = tile[r + e][c];
        if (gr + 8 <= rows && ((ld_out & 7) == 0)) {
            *(u32x4*)(out + (int64_t)gc * ld_out + gr) = *(const u32x4*)tmp;
        } else {
            for (int e = 0; e < 8; ++e)
                if (gr + e < rows) out[(int64_t)gc * ld_out + gr + e] = tmp[e];
        }
    }
}

// column sums of a [M][N] bf16 matrix added to fp32 out[N]: a workgroup owns 64 columns and every row (thread = column x one of four row
// phases, then a fixed-order sum over the phases): deterministic, no atomics
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* __restrict__ x, int64_t ld, int M, int N, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int cq = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cq;
    float s = 0.f;
    if (c < N)
        for (int r = rp; r < M; r += 4) s += bf2f(x[(int64_t)r * ld + c]);
    part[rp][cq] = s;
    __syncthreads();
    if (rp == 0 && c < N) out[c] += (part[0][cq] + part[1][cq]) + (part[2][cq] + part[3][cq]);
}

// The same sums with 16-byte loads and 128 row phases per workgroup (1024 threads = 8 column groups of 8 x 128 phases): the bias gradients
// of the trainable tower (M = images x 729 rows, N = 1152 / 4304 columns) read M / 128 rows per thread instead of M / 4 two-byte loads.
// Still one workgroup per 64 columns and a fixed-order sum over the phases: deterministic, no atomics, no workspace.
__global__ __launch_bounds__(1024) void colsum8_kernel(const uint16_t* __restrict__ x, int64_t ld, int M, int N, float* __restrict__ out) {
    __shared__ float part[128][65];
    const int cg = threadIdx.x & 7, rp = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + cg * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < N)                                              // N % 8 == 0: a column group is valid as a whole
        for (int r = rp; r < M; r += 128) {
            float f[8];
            unpack8(*(const u32x4*)(x + (int64_t)r * ld + c0), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += f[e];
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[rp][cg * 8 + e] = s[e];
    __syncthreads();
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && c < N) {
        float t = 0.f;
        for (int p = 0; p < 128; ++p) t += part[p][threadIdx.x];
        out[c] += t;
    }
}

}  // namespace

namespace {
// out[m][n] = bf16(sum_s part[s][m][n] (+ residual[m][n])), slices added in order s = 0 .. S-1 (fixed order: reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N, const uint16_t* __restrict__ res,
                                                            int64_t ldr, uint16_t* __restrict__ out, int64_t ldc) {
    const int nv = N >> 3;
    const int64_t total = (int64_t)M * nv, plane = (int64_t)M * N;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nv), c = (int)(i % nv) * 8;
        float v[8];
        const float* p = part + (int64_t)m * N + c;
        const f32x4 a0 = *(const f32x4*)p, a1 = *(const f32x4*)(p + 4);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
        for (int s = 1; s < S; ++s) {
            const f32x4 b0 = *(const f32x4*)(p + s * plane), b1 = *(const f32x4*)(p + s * plane + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (res) {
            float r[8];
            unpack8(*(const u32x4*)(res + (int64_t)m * ldr + c), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        *(u32x4*)(out + (int64_t)m * ldc + c) = pack8(v);
    }
}

// ---- the reduce launch carrying what FOLLOWS a split projection on the decode / prompt path (round 6: a launch costs 3 - 5 us whatever it
// does -- profiles/r6_attn_decode_phases.log -- and a layer of the wide decode step had thirteen of them).  Each form rounds the slice sums
// to bf16 exactly where splitk_reduce_kernel stores them and continues with the arithmetic of the kernel it replaces: the same bits as the
// launch sequence.
MM_DEV void splitk_sum8(const float* __restrict__ p, int S, int64_t plane, float (&v)[8]) {
    const f32x4 a0 = *(const f32x4*)p, a1 = *(const f32x4*)(p + 4);
    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    int s = 1;
    for (; s + 3 < S; s += 4) {                              // four slices' loads in flight at once; added in slice order
        f32x4 b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { b0[j] = *(const f32x4*)(p + (s + j) * plane); b1[j] = *(const f32x4*)(p + (s + j) * plane + 4); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[0] += b0[j].x; v[1] += b0[j].y; v[2] += b0[j].z; v[3] += b0[j].w; v[4] += b1[j].x; v[5] += b1[j].y; v[6] += b1[j].z; v[7] += b1[j].w;
        }
    }
    for (; s < S; ++s) {
        const f32x4 b0 = *(const f32x4*)(p + s * plane), b1 = *(const f32x4*)(p + s * plane + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
}
// out = bf16(sums + residual) and y = RMSNorm(out; w, eps): one workgroup per row, the row in registers (rmsnorm_fwd_kernel's layout,
// reduction order and roundings)
template <int VPT>
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(const float* __restrict__ part, int S, int M, int N, const uint16_t* __restrict__ res,
                                                                 int64_t ldr, uint16_t* __restrict__ out, int64_t ldc, const uint16_t* __restrict__ w,
                                                                 uint16_t* __restrict__ y, int64_t ldy, float eps) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int nv = N >> 3;
    const int64_t plane = (int64_t)M * N;
    float xv[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * 256;
        if (v < nv) {
            float t[8];
            splitk_sum8(part + row * N + v * 8, S, plane, t);
            if (res) {
                float r[8];
                unpack8(*(const u32x4*)(res + row * ldr + v * 8), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] += r[e];
            }
            const u32x4 o = pack8(t);
            *(u32x4*)(out + row * ldc + v * 8) = o;
            unpack8(o, xv[i]);                               // the norm reads the ROUNDED row, as the separate launch does
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += xv[i][e] * xv[i][e];
        }
    }
    ss = block_sum<256>(ss, red);
    const float rstd = rsqrtf(ss / (float)N + eps);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * 256;
        if (v < nv) {
            float wv[8], o[8];
            unpack8(*(const u32x4*)(w + v * 8), wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = wv[e] * round_bf(xv[i][e] * rstd);
            *(u32x4*)(y + row * ldy + v * 8) = pack8(o);
        }
    }
}
// act[m][c] = SiLU(g) * u of the bf16-rounded sums g = column c, u = column I + c (swiglu_fwd_kernel's arithmetic)
__global__ __launch_bounds__(256) void splitk_reduce_swiglu_kernel(const float* __restrict__ part, int S, int M, int I, uint16_t* __restrict__ act,
                                                                   int64_t ld_act) {
    const int iv = I >> 3, N = 2 * I;
    const int64_t total = (int64_t)M * iv, plane = (int64_t)M * N;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / iv;
        const int c = (int)(i % iv) * 8;
        float g[8], u[8], o[8];
        splitk_sum8(part + m * N + c, S, plane, g);
        splitk_sum8(part + m * N + I + c, S, plane, u);
        unpack8(pack8(g), g);
        unpack8(pack8(u), u);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = round_bf(g[e] / (1.0f + __expf(-g[e]))) * u[e];
        *(u32x4*)(act + m * ld_act + c) = pack8(o);
    }
}
// q|k|v sums of one new row per sequence: q rotated at positions[m] -> qkv[m][0 .. Hq d), rotated k and v -> cache row positions[m]
// (rope_kv_append_kernel's arithmetic on the bf16-rounded sums; the k | v columns of qkv are not written)
__global__ __launch_bounds__(256) void splitk_reduce_rope_append_kernel(const float* __restrict__ part, int S, int M, int Hq, int Hkv, int d,
                                                                        uint16_t* __restrict__ qkv, int64_t ld, const uint16_t* __restrict__ cos_t,
                                                                        const uint16_t* __restrict__ sin_t, const int32_t* __restrict__ positions,
                                                                        uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, int64_t ld_kv,
                                                                        int64_t bs_kv) {
    const int b = blockIdx.y;
    const int pos = positions[b];
    const int half = d >> 1, vph = half >> 3;
    const int H = Hq + Hkv, N = (Hq + 2 * Hkv) * d;
    const int64_t plane = (int64_t)M * N;
    const float* prow = part + (int64_t)b * N;
    uint16_t* krow = kc + (int64_t)b * bs_kv + (int64_t)pos * ld_kv;
    uint16_t* vrow = vc + (int64_t)b * bs_kv + (int64_t)pos * ld_kv;
    const int rot = H * vph, cpy = Hkv * d / 8;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rot + cpy; i += gridDim.x * 256) {
        if (i < rot) {
            const int v = i % vph, hd = i / vph;
            float x1[8], x2[8], c[8], sn[8], y1[8], y2[8];
            splitk_sum8(prow + (int64_t)hd * d + v * 8, S, plane, x1);
            splitk_sum8(prow + (int64_t)hd * d + half + v * 8, S, plane, x2);
            unpack8(pack8(x1), x1);
            unpack8(pack8(x2), x2);
            unpack8(*(const u32x4*)(cos_t + (int64_t)pos * d + v * 8), c);
            unpack8(*(const u32x4*)(sin_t + (int64_t)pos * d + v * 8), sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                y1[e] = round_bf(x1[e] * c[e]) + round_bf(-x2[e] * sn[e]);
                y2[e] = round_bf(x2[e] * c[e]) + round_bf(x1[e] * sn[e]);
            }
            const u32x4 o1 = pack8(y1), o2 = pack8(y2);
            if (hd < Hq) {
                uint16_t* p1 = qkv + (int64_t)b * ld + (int64_t)hd * d + v * 8;
                *(u32x4*)p1 = o1; *(u32x4*)(p1 + half) = o2;
            } else {
                uint16_t* kd = krow + (int64_t)(hd - Hq) * d + v * 8;
                *(u32x4*)kd = o1; *(u32x4*)(kd + half) = o2;
            }
        } else {
            const int j = i - rot;
            float t[8];
            splitk_sum8(prow + (int64_t)H * d + j * 8, S, plane, t);
            *(u32x4*)(vrow + j * 8) = pack8(t);
        }
    }
}

// slices for a prompt-pass problem: enough workgroups (64 x 128 tiles x slices) to put ~3 on every CU, at least 8 K tiles per slice
int splitk_slices(int64_t M, int64_t N, int64_t K) {
    if (K % 64 || N % 8 || M > 4096) return 1;
    const int64_t tiles = ((M + 63) / 64) * ((N + 127) / 128), nk = K / 64;
    int S = (int)std::min<int64_t>(8, (768 + tiles - 1) / tiles);
    while (S > 1 && nk / S < 8) --S;
    return S;
}
}  // namespace

extern "C" int64_t mm355_gemm_splitk_ws_floats(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int S = splitk_slices(M, N, K);
    return S > 1 ? (int64_t)S * M * N : 0;
}

// the K slices of A . B^T as fp32 partials workspace[slice][M][N]; `slices` = how many were written (the caller reduces them)
static int splitk_partials(const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int S, float* workspace,
                           int64_t workspace_floats, hipStream_t stream, int& slices) {
    if ((lda & 7) || (ldb & 7) || !mm_aligned16(A) || !mm_aligned16(B) || !workspace || !mm_aligned16(workspace) ||
        workspace_floats < (int64_t)S * M * N)
        return MM355_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = workspace; a.lda = lda; a.ldb = ldb; a.ldc = N;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = MM355_GEMM_OUT_F32;
    const int64_t nk = K / 64;
    a.kslice = (int)((nk + S - 1) / S) * 64;
    // tiles of 64 x 128 (four waves); up to 32 rows: 32 x 128 -- half the A-tile LDS and MFMAs on padding rows (the decode step of 17 - 32
    // sequences: 5.33 -> 5.10 ms; more slices, narrower or 2-wave tiles bought nothing or lost: profiles/r6_splitk_tile_sweep.log).  Every
    // output element sums its K slice in the same order under either tile: the same bits.
    constexpr int BN = 128;
    constexpr int LDS = 2 * (64 + BN) * 128;
    a.ntn = (int)((N + BN - 1) / BN);
    slices = (int)((K + a.kslice - 1) / a.kslice);
    if (M <= 32) {
        auto kern = gemm_nt_kernel<32, BN, 1, 4, true, 0, false>;
        static std::atomic<uint64_t> lds_ok{0};
        if (mm_ensure_dynamic_lds((const void*)kern, LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
        a.ntm = 1;
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.ntm * a.ntn), (unsigned)slices), dim3(256), 2 * (32 + BN) * 128, stream, a);
    } else {
        auto kern = gemm_nt_kernel<64, BN, 1, 4, true, 0, false>;
        static std::atomic<uint64_t> lds_ok{0};
        if (mm_ensure_dynamic_lds((const void*)kern, LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
        a.ntm = (int)((M + 63) / 64);
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.ntm * a.ntn), (unsigned)slices), dim3(256), LDS, stream, a);
    }
    return mm_launch_status();
}

extern "C" int mm355_gemm_splitk_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb, mm355_bf16* C, int64_t ldc,
                                      int64_t M, int64_t N, int64_t K, const mm355_bf16* residual, int64_t ldr, float* workspace,
                                      int64_t workspace_floats, void* stream) {
    (void)hipGetLastError();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MM355_EINVAL;
    const int S = splitk_slices(M, N, K);
    if (S <= 1)                                              // nothing to split: the plain kernel (same call a caller would have made)
        return mm355_gemm_bf16(A, lda, B, ldb, C, ldc, M, N, K, nullptr, residual, ldr, 0, residual ? MM355_GEMM_RESIDUAL : 0u, 0, stream);
    if ((lda & 7) || (ldb & 7) || (ldc & 7) || (residual && (ldr & 7)) || !mm_aligned16(A) || !mm_aligned16(B) || !mm_aligned16(C) ||
        (residual && !mm_aligned16(residual)) || !workspace || !mm_aligned16(workspace) || workspace_floats < (int64_t)S * M * N)
        return MM355_EINVAL;
    int slices = 0;
    const int rc = splitk_partials(A, lda, B, ldb, M, N, K, S, workspace, workspace_floats, (hipStream_t)stream, slices);
    if (rc != MM355_OK) return rc;
    const int64_t vecs = M * (N / 8);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>((vecs + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, slices, (int)M, (int)N, (const uint16_t*)residual, ldr, (uint16_t*)C, ldc);
    return mm_launch_status();
}

// ---- split projection + the launch that follows it (see the fused reduce kernels above).  A shape that would not be split runs the plain
// launch sequence inside the library: same results either way.
extern "C" int mm355_gemm_splitk_norm_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb, mm355_bf16* C, int64_t M, int64_t N,
                                           int64_t K, const mm355_bf16* residual, int64_t ldr, const mm355_bf16* norm_w, float eps, mm355_bf16* Y,
                                           float* workspace, int64_t workspace_floats, void* stream) {
    (void)hipGetLastError();
    if (!A || !B || !C || !Y || !norm_w || M <= 0 || N <= 0 || K <= 0 || (N & 7)) return MM355_EINVAL;
    if (!mm_aligned16(C) || !mm_aligned16(Y) || !mm_aligned16(norm_w) || (residual && ((ldr & 7) || !mm_aligned16(residual)))) return MM355_EINVAL;
    const int S = splitk_slices(M, N, K);
    if (S <= 1) {
        const int rc = mm355_gemm_bf16(A, lda, B, ldb, C, N, M, N, K, nullptr, residual, ldr, 0, residual ? MM355_GEMM_RESIDUAL : 0u, 0, stream);
        return rc != MM355_OK ? rc : mm355_rmsnorm_fwd(C, norm_w, Y, M, N, eps, stream);
    }
    const int nv = (int)(N >> 3);
    if (nv > 8 * 256) return MM355_EUNSUPPORTED;            // (the row lives in registers: mm355_rmsnorm_fwd's own limit)
    int slices = 0;
    const int rc = splitk_partials(A, lda, B, ldb, M, N, K, S, workspace, workspace_floats, (hipStream_t)stream, slices);
    if (rc != MM355_OK) return rc;
#define RN(VPT) hipLaunchKernelGGL((splitk_reduce_norm_kernel<VPT>), dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, slices, \
                                   (int)M, (int)N, (const uint16_t*)residual, ldr, (uint16_t*)C, N, (const uint16_t*)norm_w, (uint16_t*)Y, N, eps)
    if (nv <= 256) RN(1); else if (nv <= 512) RN(2); else if (nv <= 1024) RN(4); else RN(8);
#undef RN
    return mm_launch_status();
}

extern "C" int64_t mm355_gemm_splitk_swiglu_ws_floats(int64_t M, int64_t I, int64_t K) {
    if (M <= 0 || I <= 0 || K <= 0) return 0;
    const int S = splitk_slices(M, 2 * I, K);
    return S > 1 ? (int64_t)S * M * 2 * I : M * I;          // not split: the bf16 [M][2 I] gate | up rows of the plain sequence
}
extern "C" int mm355_gemm_splitk_swiglu_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wgu, int64_t ldw, mm355_bf16* act, int64_t ld_act,
                                             int64_t M, int64_t I, int64_t K, float* workspace, int64_t workspace_floats, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wgu || !act || !workspace || M <= 0 || I <= 0 || K <= 0 || (I & 7) || (ld_act & 7) || !mm_aligned16(act) || !mm_aligned16(workspace))
        return MM355_EINVAL;
    if (workspace_floats < mm355_gemm_splitk_swiglu_ws_floats(M, I, K)) return MM355_EINVAL;
    const int64_t N = 2 * I;
    const int S = splitk_slices(M, N, K);
    if (S <= 1) {
        if (ld_act != I) return MM355_EUNSUPPORTED;
        mm355_bf16* gu = (mm355_bf16*)workspace;
        const int rc = mm355_gemm_bf16(X, ldx, Wgu, ldw, gu, N, M, N, K, nullptr, nullptr, 0, 0, 0u, 0, stream);
        return rc != MM355_OK ? rc : mm355_swiglu_fwd(gu, act, M, I, stream);
    }
    int slices = 0;
    const int rc = splitk_partials(X, ldx, Wgu, ldw, M, N, K, S, workspace, workspace_floats, (hipStream_t)stream, slices);
    if (rc != MM355_OK) return rc;
    const int64_t vecs = M * (I / 8);
    hipLaunchKernelGGL(splitk_reduce_swiglu_kernel, dim3((unsigned)std::min<int64_t>((vecs + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, slices, (int)M, (int)I, (uint16_t*)act, ld_act);
    return mm_launch_status();
}

extern "C" int mm355_gemm_splitk_rope_append_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wqkv, int64_t ldw, mm355_bf16* qkv, int64_t ld_qkv,
                                                  int64_t M, int64_t Hq, int64_t Hkv, int64_t d, int64_t K, const mm355_bf16* cos_t,
                                                  const mm355_bf16* sin_t, const int32_t* positions, mm355_bf16* k_cache, mm355_bf16* v_cache,
                                                  int64_t ld_kv, int64_t batch_stride_kv, float* workspace, int64_t workspace_floats, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wqkv || !qkv || !cos_t || !sin_t || !positions || !k_cache || !v_cache || M <= 0 || Hq <= 0 || Hkv <= 0 || d <= 0 || K <= 0 ||
        (d & 15) || (ld_qkv & 7) || (ld_kv & 7) || (batch_stride_kv & 7) || M > 65535 || !mm_aligned16(qkv) || !mm_aligned16(k_cache) ||
        !mm_aligned16(v_cache) || !mm_aligned16(cos_t) || !mm_aligned16(sin_t))
        return MM355_EINVAL;
    const int64_t N = (Hq + 2 * Hkv) * d;
    const int S = splitk_slices(M, N, K);
    if (S <= 1) {
        const int rc = mm355_gemm_bf16(X, ldx, Wqkv, ldw, qkv, ld_qkv, M, N, K, nullptr, nullptr, 0, 0, 0u, 0, stream);
        return rc != MM355_OK ? rc : mm355_rope_kv_append(qkv, ld_qkv, M, Hq, Hkv, d, cos_t, sin_t, positions, k_cache, v_cache, ld_kv, batch_stride_kv, stream);
    }
    int slices = 0;
    const int rc = splitk_partials(X, ldx, Wqkv, ldw, M, N, K, S, workspace, workspace_floats, (hipStream_t)stream, slices);
    if (rc != MM355_OK) return rc;
    const int64_t work = (Hq + Hkv) * (d / 16) + Hkv * d / 8;
    hipLaunchKernelGGL(splitk_reduce_rope_append_kernel, dim3((unsigned)((work + 255) / 256), (unsigned)M), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, slices, (int)M, (int)Hq, (int)Hkv, (int)d, (uint16_t*)qkv, ld_qkv, (const uint16_t*)cos_t,
                       (const uint16_t*)sin_t, positions, (uint16_t*)k_cache, (uint16_t*)v_cache, ld_kv, batch_stride_kv);
    return mm_launch_status();
}

extern "C" int mm355_gemm_num_variants(void) { return 14; }

int mm355_gemm_st_launch(const void* args, int serialised, void* stream);   // gemm_st.hip: one wave per SIMD, hand-placed stream

#ifdef MM355_LEGACY_VARIANTS
namespace {
// the stream kernel moves whole pairs of K stages and fetches two stages ahead: K % 128 == 0, K >= 256, 31-bit tile-relative offsets
bool st_eligible(const GemmArgs& a) { return a.K >= 256 && pp_eligible(a, false, false); }
}  // namespace
#endif

extern "C" int mm355_gemm_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb, void* C, int64_t ldc,
                               int64_t M, int64_t N, int64_t K, const mm355_bf16* bias, const mm355_bf16* residual,
                               int64_t ldr, int64_t res_row_mod, uint32_t flags, int variant, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MM355_EINVAL;
    if ((K & 7) || (lda & 7) || (ldb & 7) || !mm_aligned16(A) || !mm_aligned16(B) || !mm_aligned16(C)) return MM355_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    if ((flags & MM355_GEMM_BIAS) && (!bias || !mm_aligned16(bias))) return MM355_EINVAL;
    if ((flags & MM355_GEMM_RESIDUAL) && (!residual || !mm_aligned16(residual))) return MM355_EINVAL;
    if ((flags & MM355_GEMM_GELU_ERF) && (flags & MM355_GEMM_GELU_TANH)) return MM355_EINVAL;
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.res = residual;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.res_mod = res_row_mod;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = flags; a.ntm = a.ntn = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool dma_ok = (K % 64) == 0;
    if (variant == 0) {
        // auto: LDS-DMA staging whenever K is a whole number of 64-wide tiles; the 256x256 ping-pong kernel once it
        // still yields at least one full wave of workgroups over the 256 CUs.
        const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
        const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
        if (dma_ok) {
            variant = (t256 >= 200) ? 11 : 2;                // 11 falls back to 7 when K is not a whole number of tile pairs
            // prompt-pass shapes (a few hundred rows against N = 4096 .. 6144): 128 x 128 tiles leave half of the 256 CUs without a
            // workgroup (M = 512, N = 4096: 128 tiles); 64-row tiles double the count.  Same K order per output element: same bits.
            if (variant == 2 && t128 < 224 && M > 64 && t128 * 2 >= 64) variant = 9;
        } else variant = 1;
    }
    if (!dma_ok && (variant == 2 || variant == 4 || variant >= 6)) return MM355_EUNSUPPORTED;
    switch (variant) {
        case 1: return launch_gemm<128, 128, 2, 2, false>(a, s);
        case 2: return launch_gemm<128, 128, 2, 2, true>(a, s);
        case 7: return launch_gemm<256, 256, 2, 4, true, 1>(a, s);
        case 9: return launch_gemm<64, 128, 1, 4, true>(a, s);    // 64 x 128 tiles, four waves side by side (small-M / prompt-pass shapes)
        case 11: return launch_gemm_pp(a, s);
#ifdef MM355_LEGACY_VARIANTS                                 // tools build: kernels no product path selects, kept for A/B timing
        case 3: return launch_gemm<256, 128, 4, 2, false>(a, s);
        case 4: return launch_gemm<256, 128, 4, 2, true>(a, s);
        case 5: return launch_gemm<256, 256, 2, 4, false>(a, s);
        case 6: return launch_gemm<256, 256, 2, 4, true>(a, s);
        case 8: return launch_gemm<128, 128, 2, 2, true, 1>(a, s);
        case 10: return launch_gemm_ring(a, s);
        case 12: return launch_gemm_pp2(a, s);
        case 13: case 14: return st_eligible(a) ? mm355_gemm_st_launch(&a, variant == 14, s) : MM355_EUNSUPPORTED;
#endif
#ifdef MM355_ABLATIONS                                       // TIMING-ONLY builds (tools/build_ablation.sh ... -DMM355_ABLATIONS): wrong results on purpose
        case 91: return launch_gemm_pp_t<false, false, 1>(a, s);     // no DMA
        case 92: return launch_gemm_pp_t<false, false, 2>(a, s);     // no fragment reads
        case 93: return launch_gemm_pp_t<false, false, 3>(a, s);     // every DMA issued, but always K tiles 0 / 1 (L2-resident sources)
#endif
        default: return MM355_EUNSUPPORTED;
    }
}

namespace {
// one problem of mm355_gemm_pair_bf16: plain NT form, epilogue limited to ACCUMULATE / OUT_F32
int pair_problem(GemmArgs& a, const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb, void* C, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, uint32_t flags) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MM355_EINVAL;
    if ((K & 7) || (lda & 7) || (ldb & 7) || !mm_aligned16(A) || !mm_aligned16(B) || !mm_aligned16(C)) return MM355_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    if (flags & ~(MM355_GEMM_ACCUMULATE | MM355_GEMM_OUT_F32)) return MM355_EINVAL;
    a.A = A; a.B = B; a.C = C; a.bias = nullptr; a.res = nullptr;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = 0; a.res_mod = 0;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = flags; a.ntm = a.ntn = 0;
    return pp_eligible(a, false, false) ? MM355_OK : MM355_EUNSUPPORTED;
}
}  // namespace

extern "C" int mm355_gemm_pair_bf16(const mm355_bf16* A0, int64_t lda0, const mm355_bf16* B0, int64_t ldb0, void* C0, int64_t ldc0,
                                    int64_t M0, int64_t N0, int64_t K0, uint32_t flags0,
                                    const mm355_bf16* A1, int64_t lda1, const mm355_bf16* B1, int64_t ldb1, void* C1, int64_t ldc1,
                                    int64_t M1, int64_t N1, int64_t K1, uint32_t flags1, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    GemmArgs a0, a1;
    int rc = pair_problem(a0, A0, lda0, B0, ldb0, C0, ldc0, M0, N0, K0, flags0);
    if (rc != MM355_OK) return rc;
    rc = pair_problem(a1, A1, lda1, B1, ldb1, C1, ldc1, M1, N1, K1, flags1);
    if (rc != MM355_OK) return rc;
    return launch_gemm_pp_pair(a0, a1, (hipStream_t)stream);
}

extern "C" int mm355_gemm_tn_bf16(const mm355_bf16* At, int64_t lda, const mm355_bf16* Bt, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                  int64_t N, int64_t K, uint32_t flags, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!At || !Bt || !C || M < 8 || N < 8 || K <= 0) return MM355_EINVAL;
    if ((M & 7) || (N & 7) || (lda & 7) || (ldb & 7) || !mm_aligned16(At) || !mm_aligned16(Bt) || !mm_aligned16(C)) return MM355_EINVAL;
    if (K % 64) return MM355_EUNSUPPORTED;                  // contraction rows are DMA'd unmasked: whole 64-row tiles only
    if (flags & ~(MM355_GEMM_ACCUMULATE | MM355_GEMM_OUT_F32)) return MM355_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    GemmArgs a = {};
    a.A = At; a.B = Bt; a.C = C; a.bias = nullptr; a.res = nullptr;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = 0; a.res_mod = 0;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = flags; a.ntm = a.ntn = 0;
    if (pp_eligible(a, true, true)) return launch_gemm_pp_t<true, true>(a, (hipStream_t)stream);
    return launch_gemm<256, 256, 2, 4, true, 1, true>(a, (hipStream_t)stream);
}

extern "C" int mm355_gemm_nn_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* Bt, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                  int64_t N, int64_t K, const mm355_bf16* residual, int64_t ldr, uint32_t flags, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!A || !Bt || !C || M <= 0 || N < 8 || K <= 0) return MM355_EINVAL;
    if ((N & 7) || (lda & 7) || (ldb & 7) || (ldc & 7) || !mm_aligned16(A) || !mm_aligned16(Bt) || !mm_aligned16(C)) return MM355_EINVAL;
    if (flags & ~(MM355_GEMM_ACCUMULATE | MM355_GEMM_OUT_F32 | MM355_GEMM_RESIDUAL)) return MM355_EINVAL;
    if ((flags & MM355_GEMM_RESIDUAL) && (!residual || !mm_aligned16(residual) || (ldr & 7))) return MM355_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    GemmArgs a = {};
    a.A = A; a.B = Bt; a.C = C; a.bias = nullptr; a.res = residual;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.res_mod = 0;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = flags; a.ntm = a.ntn = 0;
    if (!pp_eligible(a, false, true)) return MM355_EUNSUPPORTED;   // caller: mm355_transpose_bf16 + mm355_gemm_bf16
    return launch_gemm_pp_t<false, true>(a, (hipStream_t)stream);
}

extern "C" int mm355_transpose_bf16(const mm355_bf16* in, int64_t ld_in, int64_t rows, int64_t cols, mm355_bf16* out,
                                    int64_t ld_out, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!in || !out || rows <= 0 || cols <= 0 || (ld_in & 7) || !mm_aligned16(in) || !mm_aligned16(out)) return MM355_EINVAL;
    if (ld_out < rows || rows > 0x7fffffff || cols > 0x7fffffff) return MM355_EINVAL;   // rows of `out` would overlap / run past it
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ld_in, (int)rows, (int)cols, out, ld_out);
    return mm_launch_status();
}

extern "C" int mm355_gemm_swiglu_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wgu, int64_t ldw, mm355_bf16* gu, int64_t ld_gu,
                                      mm355_bf16* act, int64_t ld_act, int64_t M, int64_t I, int64_t K, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wgu || !gu || !act || M <= 0 || I <= 0 || K <= 0) return MM355_EINVAL;
    if ((ldx & 7) || (ldw & 7) || (ld_gu & 7) || (ld_act & 7) || !mm_aligned16(X) || !mm_aligned16(Wgu) || !mm_aligned16(gu) || !mm_aligned16(act))
        return MM355_EINVAL;
    if (M > 0x7fffffff || I > 0x3fffffff || K > 0x7fffffff) return MM355_EINVAL;
    // whole 128-channel tiles, whole pairs of K tiles, 31-bit weight offsets (2 I rows from the weight's first row)
    if ((I & 127) || K < 128 || (K & 127) || (2 * I * ldw + K) * 2 >= 0x7fffffffLL || (256 * ldx + K) * 2 >= 0x7fffffffLL) return MM355_EUNSUPPORTED;
    GemmArgs a = {};
    a.A = X; a.B = Wgu; a.C = gu; a.bias = nullptr; a.res = act;
    a.lda = ldx; a.ldb = ldw; a.ldc = ld_gu; a.ldr = ld_act; a.res_mod = I;
    a.M = (int)M; a.N = (int)(2 * I); a.K = (int)K; a.flags = 0; a.ntm = a.ntn = 0;
    static std::atomic<uint64_t> lds_ok{0};
    if (mm_ensure_dynamic_lds((const void*)gemm_pp_swiglu_kernel, PP_LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    const int64_t total = pp_prepare(a);
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(gemm_pp_swiglu_kernel, dim3((unsigned)total), dim3(512), PP_LDS, (hipStream_t)stream, a);
    return mm_launch_status();
}

extern "C" int mm355_gemm_rope_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wqkv, int64_t ldw, mm355_bf16* qkv, int64_t ld_qkv,
                                    const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* pos_offset,
                                    int64_t M, int64_t N, int64_t K, int64_t L, int64_t n_rot, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wqkv || !qkv || !cos_t || !sin_t || M <= 0 || N <= 0 || K <= 0 || L <= 0 || n_rot < 0 || n_rot > N) return MM355_EINVAL;
    if ((ldx & 7) || (ldw & 7) || (ld_qkv & 7) || !mm_aligned16(X) || !mm_aligned16(Wqkv) || !mm_aligned16(qkv) || !mm_aligned16(cos_t) || !mm_aligned16(sin_t))
        return MM355_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff || L > 0x7fffffff) return MM355_EINVAL;
    // heads of 128 in whole 256-column tiles, whole pairs of K tiles, 31-bit tile-relative offsets
    if ((N & 255) || (n_rot & 127) || K < 128 || (K & 127) || (256 * ldx + K) * 2 >= 0x7fffffffLL || (256 * ldw + K) * 2 >= 0x7fffffffLL) return MM355_EUNSUPPORTED;
    GemmArgs a = {};
    a.A = X; a.B = Wqkv; a.C = qkv; a.bias = (const uint16_t*)pos_offset; a.res = nullptr;
    a.lda = ldx; a.ldb = ldw; a.ldc = ld_qkv; a.ldr = 0; a.res_mod = n_rot;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = 0; a.ntm = a.ntn = 0;
    a.aux0 = (uint16_t*)cos_t; a.aux1 = (uint16_t*)sin_t; a.ld_aux = L;
    static std::atomic<uint64_t> lds_ok{0};
    if (mm_ensure_dynamic_lds((const void*)gemm_pp_rope_kernel, PP_LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    const int64_t total = pp_prepare(a);
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(gemm_pp_rope_kernel, dim3((unsigned)total), dim3(512), PP_LDS, (hipStream_t)stream, a);
    return mm_launch_status();
}

extern "C" int mm355_gemm_swiglu_bwd_bf16(const mm355_bf16* dY, int64_t ldy, const mm355_bf16* WdT, int64_t ldw, const mm355_bf16* gu, int64_t ld_gu,
                                          mm355_bf16* dgu, int64_t ld_dgu, mm355_bf16* actT, mm355_bf16* dguT, int64_t ldT,
                                          int64_t M, int64_t I, int64_t K, void* stream) {
    (void)hipGetLastError();
    if (!dY || !WdT || !gu || !dgu || !actT || !dguT || M <= 0 || I <= 0 || K <= 0) return MM355_EINVAL;
    if ((ldy & 7) || (ldw & 7) || (ld_gu & 7) || (ld_dgu & 7) || (ldT & 7) || ldT < M || !mm_aligned16(dY) || !mm_aligned16(WdT) || !mm_aligned16(gu)
        || !mm_aligned16(dgu) || !mm_aligned16(actT) || !mm_aligned16(dguT))
        return MM355_EINVAL;
    if (M > 0x7fffffff || I > 0x3fffffff || K > 0x7fffffff) return MM355_EINVAL;
    // whole 64-channel wave columns, 8-row vectors, whole pairs of K tiles, 31-bit tile-relative operand offsets
    if ((I & 63) || (M & 7) || K < 128 || (K & 127) || (256 * ldy + K) * 2 >= 0x7fffffffLL || (256 * ldw + K) * 2 >= 0x7fffffffLL) return MM355_EUNSUPPORTED;
    GemmArgs a = {};
    a.A = dY; a.B = WdT; a.C = dgu; a.bias = nullptr; a.res = gu;
    a.lda = ldy; a.ldb = ldw; a.ldc = ld_dgu; a.ldr = ld_gu; a.res_mod = 0;
    a.M = (int)M; a.N = (int)I; a.K = (int)K; a.flags = 0; a.ntm = a.ntn = 0;
    a.aux0 = actT; a.aux1 = dguT; a.ld_aux = ldT;
    static std::atomic<uint64_t> lds_ok{0};
    if (mm_ensure_dynamic_lds((const void*)gemm_pp_swiglu_bwd_kernel, SWB_LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    const int64_t total = pp_prepare(a);
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(gemm_pp_swiglu_bwd_kernel, dim3((unsigned)total), dim3(512), SWB_LDS, (hipStream_t)stream, a);
    return mm_launch_status();
}

extern "C" int mm355_colsum_bf16(const mm355_bf16* dY, int64_t ld, int64_t M, int64_t N, float* db_f32, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!dY || !db_f32 || M <= 0 || N <= 0) return MM355_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff) return MM355_EINVAL;
    if (!(N & 7) && !(ld & 7) && mm_aligned16(dY) && M >= 256)
        hipLaunchKernelGGL(colsum8_kernel, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, dY, ld, (int)M, (int)N, db_f32);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dY, ld, (int)M, (int)N, db_f32);
    return mm_launch_status();
}
