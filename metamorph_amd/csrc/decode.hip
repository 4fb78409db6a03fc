// Decode-shape kernels (gfx950): the same decoder as the training path, but for a handful of new rows per step against a
// KV cache (greedy_decode / generate, reference metamorph_llama.py:502-597, 665-717, which re-runs the whole
// prefix every step with use_cache=False).  Everything here is HBM-bound streaming, not MFMA work:
//   * gemv_*_kernel    y[M,N] = x[M,K] W[N,K]^T for M <= 16: every weight row is read exactly once, 16 B per lane, four rows per
//                      wave in flight, fp32 accumulation, the usual bias / GELU / residual epilogue (or SwiGLU / RoPE + cache append)
//   * attn_decode_*    one query row per (sample, head) against [kv_len] cached keys: one workgroup per 1024 cached rows (a cache bound of
//                      <= 1024 rows: per query head, no merge), partial (max, sum, o) per group merged by the group that arrives last
#include "mm355_common.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------------------ GEMV
// KS = 1: a wave owns R = 4 weight rows over the whole K.  KS = 2 / 4 (small N): the 4 waves of a workgroup form 4 / KS row
// groups x KS K-slices, partial sums meet in LDS -- 2x / 4x more waves in flight for the same N.
template <int MR, int KS>
__global__ __launch_bounds__(NT) void gemv_kernel(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ W, int64_t ldw,
                                                  void* __restrict__ y, int64_t ldy, int M, int N, int K, const uint16_t* __restrict__ bias,
                                                  const uint16_t* __restrict__ res, int64_t ldr, uint32_t flags) {
    constexpr int R = 4;                                     // weight rows per wave
    constexpr int RG = (NT / 64) / KS;                       // row groups per workgroup
    __shared__ float part[NT / 64][MR * R];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = wave / KS, ks = wave % KS;
    const int n0 = (blockIdx.x * RG + rg) * R;
    float acc[MR][R];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
    const uint16_t* wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = W + (int64_t)min(n0 + r, N - 1) * ldw;
    auto fma_chunk = [&](const u32x4 (&wv)[R], int k) {
        float xf[MR][8];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < M) unpack8(*(const u32x4*)(x + (int64_t)m * ldx + k), xf[m]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[m][e] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float wf[8];
            unpack8(wv[r], wf);
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[m][r] = fmaf(xf[m][e], wf[e], acc[m][r]);
        }
    };
    // this wave's K slice in 512-element chunks, two chunks per trip: eight 16-B weight loads per lane in flight
    const int nch = (K + 511) / 512;
    const int c0 = (nch * ks) / KS, c1 = (nch * (ks + 1)) / KS;
    int c = c0;
    if (n0 < N) {
        for (; c + 1 < c1; c += 2) {
            const int k = c * 512 + lane * 8;
            u32x4 w0[R], w1[R];
            const bool in1 = k + 512 < K;
#pragma unroll
            for (int r = 0; r < R; ++r) { w0[r] = *(const u32x4*)(wr[r] + k); w1[r] = in1 ? *(const u32x4*)(wr[r] + k + 512) : u32x4{0u, 0u, 0u, 0u}; }
            fma_chunk(w0, k);
            if (in1) fma_chunk(w1, k + 512);
        }
        for (; c < c1; ++c) {
            const int k = c * 512 + lane * 8;
            if (k < K) {
                u32x4 w0[R];
#pragma unroll
                for (int r = 0; r < R; ++r) w0[r] = *(const u32x4*)(wr[r] + k);
                fma_chunk(w0, k);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = wave_sum(acc[m][r]);
    if constexpr (KS > 1) {
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int r = 0; r < R; ++r) part[wave][m * R + r] = acc[m][r];
        }
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < KS; ++q) v += part[rg * KS + q][m * R + r];
                acc[m][r] = v;
            }
    }
    if (n0 >= N) return;
    // lane (m * R + r) finishes output (m, n0 + r)
    if (lane < MR * R) {
        const int m = lane / R, r = lane % R, n = n0 + r;
        if (m < M && n < N) {
            float v = 0.f;
#pragma unroll
            for (int mm = 0; mm < MR; ++mm)
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                    if (mm == m && rr == r) v = acc[mm][rr];
            if (flags & MM355_GEMM_BIAS) v += bf2f(bias[n]);
            if (flags & MM355_GEMM_GELU_ERF) v = gelu_erf_f(v);
            if (flags & MM355_GEMM_GELU_TANH) v = gelu_tanh_f(v);
            if (flags & MM355_GEMM_RESIDUAL) v += bf2f(res[(int64_t)m * ldr + n]);
            if (flags & MM355_GEMM_OUT_F32) ((float*)y)[(int64_t)m * ldy + n] = v;
            else ((uint16_t*)y)[(int64_t)m * ldy + n] = f2bf(v);
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused decode GEMVs
// The decode step of one layer is nine tiny launches around four weight streams; three of them fold into the GEMVs that consume /
// produce their rows (round 4):
//   PRENORM  the RMSNorm in front of the qkv and gate|up projections: every workgroup reduces the input rows itself (8 KiB each,
//            L2-resident) while its first weight loads are in flight, and forms bf16(w * bf16(x * rstd)) -- rmsnorm_fwd_kernel's
//            arithmetic and reduction order, bit for bit -- as the x operand;
//   MODE 1   SiLU(gate) * up in the epilogue: a wave owns gate rows c, c + 1 AND up rows I + c, I + 1 + c of the fused weight, rounds
//            both to bf16 as the unfused GEMV would have stored them and applies swiglu_fwd_kernel's arithmetic; gu never exists;
//   MODE 2   RoPE + KV-cache append in the epilogue: a wave owns the rotation partners j, j + 1, j + d/2, j + 1 + d/2 of one head (v rows:
//            four neighbours), rotates the bf16-rounded q / k values at the DEVICE-side position as rope_kv_append_kernel does and
//            writes q to the row buffer, k / v to the cache row.
// One or two rows: gemv_deep_kernel; 3 .. 16: gemv_mfma_kernel (both below; the modes are template parameters of either).
struct GemvFusedArgs {
    const uint16_t* x; int64_t ldx;
    const uint16_t* W; int64_t ldw;
    int M, N, K;                                             // N = weight rows
    const uint16_t* norm_w; float eps;                       // PRENORM
    uint16_t* out; int64_t ld_out;                           // MODE 1: act [M][I]; MODE 2: qkv row buffer [M][N]
    int I;                                                   // MODE 1
    int Hq, Hkv, d;                                          // MODE 2
    const uint16_t* cos_t; const uint16_t* sin_t; const int32_t* positions;
    uint16_t* kc; uint16_t* vc; int64_t ld_kv, bs_kv;
};

// ------------------------------------------------------------------------------------------------ 5 .. 16 rows: the same GEMVs on MFMA
// Round 5 (batched decode: all rows of a batch / all beams go through the layers in ONE pass).  The vector-ALU form (gemv_deep_kernel below)
// holds M x 4 accumulators and M x-vectors per lane and reads M x 16 bytes of LDS per chunk: fine up to four rows, LDS- and
// register-bound at eight (gate|up 47 us).  Here the dot products run on v_mfma_f32_16x16x32_bf16 and the kernel stays a weight stream at
// any M <= 16 (gate|up 44.6 us at eight rows, 52 at sixteen):
//   * a wave owns FOUR units (the 4-row units of the kernels above: plain = rows 4u .. 4u+3; SwiGLU = gate rows c, c+1 and up rows I+c,
//     I+c+1; RoPE = the rotation partners j, j+1, j+d/2, j+1+d/2 of one head) = 16 weight rows as the A operand (lane fr = lane & 15,
//     fq = lane >> 4 holds row fr at k + 8 fq).  WLDS (the product form): a block of 256 columns is loaded COALESCED -- instruction u =
//     rows 2u and 2u + 1, 512 contiguous bytes each, eight instructions per block, the next block issued before the MFMAs of the current
//     one -- and re-laid out into fragments through 8.5 KiB of wave-private LDS (eight ds_write_b128 + eight ds_read_b128 per block, rows
//     544 B apart: conflict-free; a wave's LDS operations execute in order, so no barrier).  The first form loaded straight into the
//     fragment layout, i.e. 64 contiguous bytes per row and instruction: gate|up at 16 rows 69 us against 53 now, lm_head 275 -> 204
//     (it remains for the folded-norm variant, whose x rows take the LDS);
//   * the x rows are the B operand (row m = fr, zero beyond M): XK (the product form wherever it wins, see the launcher): through
//     double-buffered LDS windows of one block, staged once per workgroup; else from L2 as they are, or -- PRENORM -- from LDS, where every workgroup
//     forms bf16(w * bf16(x * rstd)) once (rmsnorm_fwd_kernel's arithmetic and reduction order);
//   * D[16 weight rows][16 x rows]: lane (m = fr, unit fq) ends with the FOUR outputs of one unit for one x row -- exactly what the
//     epilogues of the kernels above take (bias / GELU / residual; SiLU(g) u; RoPE + cache append);
//   * K is split over KS = 1 / 2 / 4 waves of a workgroup by the number of units (>= ~1000 waves in flight), partial tiles meet in LDS in
//     a fixed order.  KS depends on (weight rows, K) only, so a fused kernel and the launch sequence it replaces see the same sums.
struct GemvMfmaArgs {
    GemvFusedArgs f;                                         // x, W, M, N, K, norm_w / eps, out (MODE 1 / 2) and the MODE fields
    void* y; int64_t ldy;                                    // MODE 0
    const uint16_t* bias; const uint16_t* res; int64_t ldr; uint32_t flags;
    int ks;                                                  // waves per unit group (1, 2, 4)
    int xs_stride;                                           // PRENORM: LDS row stride of the normalised x rows, bytes
};

template <int MODE>
MM_DEV void unit_rows(const GemvFusedArgs& a, int unit, int (&rows)[4]) {
    if constexpr (MODE == 0) {
        rows[0] = unit * 4; rows[1] = rows[0] + 1; rows[2] = rows[0] + 2; rows[3] = rows[0] + 3;
    } else if constexpr (MODE == 1) {
        const int c = unit * 2;
        rows[0] = c; rows[1] = c + 1; rows[2] = a.I + c; rows[3] = a.I + c + 1;
    } else {
        const int upd = a.d / 4, nrot = (a.Hq + a.Hkv) * upd;
        if (unit < nrot) {
            const int hd = unit / upd, j = (unit % upd) * 2;
            rows[0] = hd * a.d + j; rows[1] = rows[0] + 1; rows[2] = rows[0] + a.d / 2; rows[3] = rows[2] + 1;
        } else {
            const int b0 = (a.Hq + a.Hkv) * a.d + (unit - nrot) * 4;
            rows[0] = b0; rows[1] = b0 + 1; rows[2] = b0 + 2; rows[3] = b0 + 3;
        }
    }
}

// RG = 2 (round 6, 17 .. 32 x rows on the shapes whose waves share one K range, XK == 1: gate|up, lm_head): a SECOND group of 16 x rows rides
// on the same weight fragments -- every fragment read back from the re-layout buffer feeds two MFMAs, the x windows hold 32 rows (35 KB),
// the weights are streamed once.  (Sharing weights between WORKGROUPS through L2 was measured and lost: profiles/r6_gemv_row_groups.log.)
// Each x row's column of D is computed exactly as in a 16-row call: the bits of the 16-row calls on the row slices.
template <int MODE, bool PRENORM, int GR, bool WLDS = false, int XK = 0, int RG = 1>
__global__ __launch_bounds__(NT) void gemv_mfma_kernel(GemvMfmaArgs g) {
    static_assert(!WLDS || GR == 1, "the LDS re-layout is written for one group of 16 rows per wave");
    static_assert(XK == 0 || (WLDS && !PRENORM), "x windows come with the coalesced weight stream");
    static_assert(RG == 1 || (RG == 2 && XK == 1 && GR == 1), "two x row groups: one K range per workgroup, x windows in LDS");
    const GemvFusedArgs& a = g.f;
    constexpr int WROW = 512 + 32;                           // WLDS: bytes per weight row of a 256-column block in LDS (+ 32: every 16-lane group the hardware services together
                                                             // reads 16 different 16-byte slots of the 256-byte bank row; + 16 leaves two-way conflicts: tests/test_host_logic.py)
    __shared__ __attribute__((aligned(16))) unsigned char wl[WLDS ? NT / 64 : 1][WLDS ? 16 * WROW : 16];
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];          // PRENORM: [M][xs_stride] normalised x rows (bf16)
    __shared__ float part[RG == 1 ? NT / 64 : 1][GR][64][4];
    __shared__ float red[NT / 64];
    __shared__ float rstd_s[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = a.K, M = a.M, KS = XK ? XK : g.ks;
    // a wave owns GR groups of 16 weight rows (= 4 GR units) over its K slice: one x fragment feeds GR MFMAs
    const int grp0 = (blockIdx.x * ((NT / 64) / KS) + wave / KS) * GR, ks = wave % KS;
    int rows[4];
    const uint16_t* wp[GR];
    uint32_t wo[GR];                                         // the same rows as byte offsets into the weight buffer (host: N * ldw * 2 < 4 GiB)
#pragma unroll
    for (int gi = 0; gi < GR; ++gi) {                        // A operand: this lane's weight row = row (fr & 3) of unit (grp0 + gi) * 4 + (fr >> 2)
        unit_rows<MODE>(a, (grp0 + gi) * 4 + (fr >> 2), rows);
        int myrow = rows[0];
#pragma unroll
        for (int r = 1; r < 4; ++r) myrow = (fr & 3) == r ? rows[r] : myrow;
        wp[gi] = a.W + (int64_t)min(max(myrow, 0), a.N - 1) * a.ldw + fq * 8;
        wo[gi] = (uint32_t)min(max(myrow, 0), a.N - 1) * (uint32_t)a.ldw * 2u + (uint32_t)fq * 16u;
    }
    // WLDS: the same 16 rows loaded COALESCED -- instruction u = rows 2u (lanes 0..31) and 2u + 1 (lanes 32..63), 512 contiguous bytes
    // each -- and re-laid out into the fragment form through wave-private LDS
    uint32_t woc[8];
    if constexpr (WLDS) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ri = 2 * u + (lane >> 5);
            unit_rows<MODE>(a, grp0 * 4 + (ri >> 2), rows);
            int row = rows[0];
#pragma unroll
            for (int r = 1; r < 4; ++r) row = (ri & 3) == r ? rows[r] : row;
            woc[u] = (uint32_t)min(max(row, 0), a.N - 1) * (uint32_t)a.ldw * 2u + (uint32_t)(lane & 31) * 16u;
        }
    }
    const uint16_t* xp = a.x + (int64_t)min(fr, M - 1) * a.ldx + fq * 8;
    const uint16_t* xp1 = a.x + (int64_t)min(16 + fr, M - 1) * a.ldx + fq * 8;      // RG == 2: the lane's row of the second group
    const uint32_t xo = (uint32_t)min(fr, M - 1) * (uint32_t)a.ldx * 2u + (uint32_t)fq * 16u;
    // buffer descriptors: a load whose offset lies beyond num_records returns zeros WITHOUT touching memory -- the branch-free way to
    // skip the prefetch behind the last block (a branch around a prefetch makes hipcc wait for it at the merge: no pipelining)
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (uint32_t)((uint64_t)(a.N - 1) * a.ldw * 2 + (uint64_t)K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (uint32_t)((uint64_t)(M - 1) * a.ldx * 2 + (uint64_t)K * 2), 0x00020000);
    const uint32_t OOB = 0xf0000000u;
    const int nst = (K + 31) >> 5;                            // k-steps of 32
    const int s0 = (nst * ks) / KS, s1 = (nst * (ks + 1)) / KS;
    constexpr int U = 8 / GR;                                // k-steps per block: GR x U = 8 weight fragments per buffer, two buffers
    u32x4 wa[GR][U], wb[GR][U], xa[U], xb[U];
    const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
    // The main loop runs over this slice's COMPLETE blocks of U k-steps with plain loads -- no predicate, no select behind a load (a
    // select waits for its load on the spot and serialises the stream); what is left (< U steps, and the one step that is partial in k when
    // K % 32 != 0) goes through a step-at-a-time tail.  x rows beyond M are read from row M - 1: they only feed D columns nobody stores.
    // The x fragments are PREFETCHED like the weights, one block ahead and issued BEFORE the weight loads of that block: the vector memory
    // counter retires in order.  PRENORM reads them from LDS (its own counter).
    const int e1 = min(s1, (K & 31) ? nst - 1 : nst);        // end of the steps that are complete in k
    const int nb = max(e1 - s0, 0) / U;
    auto loadw = [&](u32x4 (&w)[GR][U], int s, uint32_t skip) {   // skip = 0 | OOB (wave-uniform)
        if constexpr (WLDS) {
#pragma unroll
            for (int u = 0; u < U; ++u) w[0][u] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (woc[u] + (uint32_t)s * 64u) | skip, 0, 2);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int gi = 0; gi < GR; ++gi) w[gi][u] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (wo[gi] + (uint32_t)(s + u) * 64u) | skip, 0, 2);
        }
    };
    auto loadx = [&](u32x4 (&xv)[U], int s, uint32_t skip) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (PRENORM) xv[u] = *(const u32x4*)(xs + min(fr, M - 1) * g.xs_stride + (min(s + u, nst - 1) * 32 + fq * 8) * 2);
            else xv[u] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (xo + (uint32_t)(s + u) * 64u) | skip, 0, 0);
        }
    };
    // XK (the product form for 5 .. 16 rows): the x rows of a block go through LDS once per WORKGROUP instead of once per wave from L2 --
    // at 16 rows the x fragments were as many bytes as the weights.  Two buffers of XK slices (one per K slice of the workgroup's waves) x
    // 16 rows x 256 columns, rows XROW bytes apart (the fragment reads are those of the weight re-layout: conflict-free); every thread
    // stages 2 XK vectors per block, loaded two blocks ahead and stored behind the barrier that retires the buffer's previous block.  All
    // waves run the same number of block pairs (nbu: the longest slice), a wave past its own slice multiplies zero weights.
    constexpr int XROW = 512 + 32, XVN = XK ? 2 * XK * RG : 1;
    uint32_t xsb[XVN], xld[XVN];
    int xnb[XVN];
    int nbu = nb;
    if constexpr (XK > 0) {
        const int complete = (K & 31) ? nst - 1 : nst;
        nbu = 0;
#pragma unroll
        for (int q = 0; q < XK; ++q) nbu = max(nbu, max(min((nst * (q + 1)) / XK, complete) - (nst * q) / XK, 0) / U);
#pragma unroll
        for (int i = 0; i < XVN; ++i) {
            const int vi = threadIdx.x + NT * i, qr = vi >> 9, q = qr / RG, m = (qr % RG) * 16 + ((vi >> 5) & 15), c = vi & 31;
            const int q0 = (nst * q) / XK, q1 = min((nst * (q + 1)) / XK, complete);
            xnb[i] = m < M ? max(q1 - q0, 0) / U : 0;          // (a row beyond M: the load is skipped and returns zeros, which are stored)
            xsb[i] = (uint32_t)min(m, M - 1) * (uint32_t)a.ldx * 2u + (uint32_t)(q0 * 32 + c * 8) * 2u;
            xld[i] = (uint32_t)((q * 16 * RG + m) * XROW + c * 16);
        }
    }
    auto xstage_load = [&](u32x4 (&r)[XVN], int b) {
#pragma unroll
        for (int i = 0; i < XVN; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (xsb[i] + (uint32_t)b * 512u) | (b < xnb[i] ? 0u : OOB), 0, 0);
    };
    auto xstage_store = [&](const u32x4 (&r)[XVN], int b) {
#pragma unroll
        for (int i = 0; i < XVN; ++i) *(u32x4*)(xs + (size_t)(b & 1) * XK * 16 * RG * XROW + xld[i]) = r[i];
    };
    auto xfrag = [&](u32x4 (&xv)[U], int b, int rg = 0) {
        const unsigned char* base = xs + (size_t)((b & 1) * XK + ks) * 16 * RG * XROW + (RG == 1 ? min(fr, M - 1) : rg * 16 + fr) * XROW + fq * 16;
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = *(const u32x4*)(base + u * 64);
    };
    u32x4 ra[XVN], rb[XVN];
    if constexpr (XK > 0) {                                  // x first into the in-order vector-memory queue
        xstage_load(ra, 0);
        xstage_load(rb, 1);
    }
    loadw(wa, s0, nb > 0 ? 0u : OOB);                        // the weight stream starts before the norm reduction
    if constexpr (XK > 0) {
        xstage_store(ra, 0);
        xstage_store(rb, 1);
        __syncthreads();
    }
    if constexpr (PRENORM) {
        const int nv = K >> 3;
        for (int m = 0; m < M; ++m) {                        // rmsnorm_fwd_kernel's reduction: thread t sums elements 8 (t + 256 i) .. + 7 in order
            float ss = 0.f;
            for (int v = threadIdx.x; v < nv; v += NT) {
                float xv[8];
                unpack8(*(const u32x4*)(a.x + (int64_t)m * a.ldx + v * 8), xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += xv[e] * xv[e];
            }
            ss = block_sum<NT>(ss, red);
            if (threadIdx.x == 0) rstd_s[m] = rsqrtf(ss / (float)K + a.eps);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < M * nv; i += NT) {
            const int m = i / nv, v = i % nv;
            float xv[8], nw[8];
            unpack8(*(const u32x4*)(a.x + (int64_t)m * a.ldx + v * 8), xv);
            unpack8(*(const u32x4*)(a.norm_w + v * 8), nw);
            const float rs = rstd_s[m];
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = nw[e] * round_bf(xv[e] * rs);
            *(u32x4*)(xs + (int64_t)m * g.xs_stride + v * 16) = pack8(xv);
        }
        // the columns between K and the end of its last 32-column step: a prefetched block behind the slice reads them against zero
        // weights, and 0 x (whatever the previous kernel left in LDS) may be NaN
        const int npad = ((K + 31) >> 5) * 4 - nv;
        for (int i = threadIdx.x; i < M * npad; i += NT)
            *(u32x4*)(xs + (int64_t)(i / npad) * g.xs_stride + (nv + i % npad) * 16) = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    f32x4 acc[GR][2], acc1[2];                               // acc1: the second x row group (RG == 2)
#pragma unroll
    for (int gi = 0; gi < GR; ++gi) acc[gi][0] = acc[gi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[0] = acc1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    int mm_block = 0;                                        // RG == 2: the block whose second-group x fragments mm() reads itself
    auto mm = [&](const u32x4 (&w)[GR][U], const u32x4 (&xv)[U]) {
        if constexpr (WLDS) {
            unsigned char* my = wl[wave];                    // (one wave's LDS operations execute in order: no barrier)
#pragma unroll
            for (int u = 0; u < U; ++u) *(u32x4*)(my + (2 * u + (lane >> 5)) * WROW + (lane & 31) * 16) = w[0][u];
            u32x4 fa[U];
#pragma unroll
            for (int u = 0; u < U; ++u) fa[u] = *(const u32x4*)(my + fr * WROW + u * 64 + fq * 16);
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc[0][u & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[u]), __builtin_bit_cast(bf16x8, xv[u]), acc[0][u & 1], 0, 0, 0);
            if constexpr (RG == 2) {                          // the same weight fragments against x rows 16 .. 31
                u32x4 x1[U];
                xfrag(x1, mm_block, 1);
#pragma unroll
                for (int u = 0; u < U; ++u)
                    acc1[u & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[u]), __builtin_bit_cast(bf16x8, x1[u]), acc1[u & 1], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)                          // 2 GR independent chains: no MFMA waits for the one before it
#pragma unroll
            for (int gi = 0; gi < GR; ++gi)
                acc[gi][u & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[gi][u]), __builtin_bit_cast(bf16x8, xv[u]),
                                                                         acc[gi][u & 1], 0, 0, 0);
    };
    if constexpr (XK > 0) {
        for (int bk = 0; bk < nbu; bk += 2) {
            const int s = s0 + bk * U;
            const uint32_t skb = bk + 1 < nb ? 0u : OOB, ska = bk + 2 < nb ? 0u : OOB;
            xstage_load(ra, bk + 2);
            xstage_load(rb, bk + 3);
            loadw(wb, s + U, skb);
            __builtin_amdgcn_sched_barrier(0);
            xfrag(xa, bk);
            mm_block = bk;
            mm(wa, xa);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                 // every wave is done with buffer 0's block
            xstage_store(ra, bk + 2);
            loadw(wa, s + 2 * U, ska);
            __builtin_amdgcn_sched_barrier(0);
            xfrag(xb, bk + 1);
            mm_block = bk + 1;
            mm(wb, xb);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                 // ... with buffer 1's; and block bk + 2 is in buffer 0 for everyone
            xstage_store(rb, bk + 3);
        }
    } else {
    loadx(xa, s0, nb > 0 ? 0u : OOB);
    for (int bk = 0; bk < nb; bk += 2) {                     // (sched_barrier: hipcc otherwise sinks the prefetch loads behind the MFMAs that do not need them)
        const int s = s0 + bk * U;
        const uint32_t skb = bk + 1 < nb ? 0u : OOB, ska = bk + 2 < nb ? 0u : OOB;
        loadx(xb, s + U, skb);
        loadw(wb, s + U, skb);
        __builtin_amdgcn_sched_barrier(0);
        mm(wa, xa);
        __builtin_amdgcn_sched_barrier(0);
        loadx(xa, s + 2 * U, ska);
        loadw(wa, s + 2 * U, ska);
        __builtin_amdgcn_sched_barrier(0);
        mm(wb, xb);                                          // (a skipped block is all zeros: adds nothing)
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    for (int st = s0 + nb * U; st < s1; ++st) {              // the tail: one step at a time, the k tail selected to zero
        const bool in = st * 32 + fq * 8 < K;
        const int off = in ? st * 32 : 0;
        u32x4 xv;
        if constexpr (PRENORM) xv = *(const u32x4*)(xs + min(fr, M - 1) * g.xs_stride + (off + fq * 8) * 2);
        else xv = *(const u32x4*)(xp + off);
        xv = in ? xv : z4;
#pragma unroll
        for (int gi = 0; gi < GR; ++gi) {
            const u32x4 wv = *(const u32x4*)(wp[gi] + off);
            acc[gi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xv), acc[gi][0], 0, 0, 0);
            if constexpr (RG == 2) {
                u32x4 xw = *(const u32x4*)(xp1 + off);
                xw = in ? xw : z4;
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xw), acc1[0], 0, 0, 0);
            }
        }
    }
    f32x4 dd[GR];
#pragma unroll
    for (int gi = 0; gi < GR; ++gi) dd[gi] = acc[gi][0] + acc[gi][1];
    if (KS > 1) {
#pragma unroll
        for (int gi = 0; gi < GR; ++gi)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][gi][lane][r] = dd[gi][r];
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int gi = 0; gi < GR; ++gi) {
            dd[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < KS; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) dd[gi][r] += part[wave + q][gi][lane][r];
        }
    }
    // lane (m = fr, unit fq) of group gi: D rows 4 fq .. 4 fq + 3 = the four outputs of unit (grp0 + gi) * 4 + fq for x row m
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
    const int m = rg * 16 + fr;
    if (m >= M) return;
#pragma unroll
    for (int gi = 0; gi < GR; ++gi) {
        const f32x4 d = rg == 0 ? dd[gi] : acc1[0] + acc1[1];
        unit_rows<MODE>(a, (grp0 + gi) * 4 + fq, rows);
        if constexpr (MODE == 0) {
            const uint32_t flags = g.flags;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = rows[r];
                if (n >= a.N) continue;
                float v = d[r];
                if (flags & MM355_GEMM_BIAS) v += bf2f(g.bias[n]);
                if (flags & MM355_GEMM_GELU_ERF) v = gelu_erf_f(v);
                if (flags & MM355_GEMM_GELU_TANH) v = gelu_tanh_f(v);
                if (flags & MM355_GEMM_RESIDUAL) v += bf2f(g.res[(int64_t)m * g.ldr + n]);
                if (flags & MM355_GEMM_OUT_F32) ((float*)g.y)[(int64_t)m * g.ldy + n] = v;
                else ((uint16_t*)g.y)[(int64_t)m * g.ldy + n] = f2bf(v);
            }
        } else {
            if (rows[3] >= a.N || (MODE == 1 && rows[1] >= a.I)) continue;
            float v4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = round_bf(d[r]);  // what the unfused GEMV stores
            if constexpr (MODE == 1) {
                float o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) o[e] = round_bf(v4[e] / (1.0f + __expf(-v4[e]))) * v4[2 + e];
                *(uint32_t*)(a.out + (int64_t)m * a.ld_out + rows[0]) = pack2bf(o[0], o[1]);
            } else {
                const int nqk = (a.Hq + a.Hkv) * a.d;
                const int pos = a.positions[m];
                if (rows[0] < nqk) {
                    const int hd = rows[0] / a.d, j = rows[0] % a.d;
                    float y1[2], y2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float c = bf2f(a.cos_t[(int64_t)pos * a.d + j + e]), sn = bf2f(a.sin_t[(int64_t)pos * a.d + j + e]);
                        y1[e] = round_bf(v4[e] * c) + round_bf(-v4[2 + e] * sn);
                        y2[e] = round_bf(v4[2 + e] * c) + round_bf(v4[e] * sn);
                    }
                    uint16_t* dst = hd < a.Hq ? a.out + (int64_t)m * a.ld_out + rows[0]
                                              : a.kc + (int64_t)m * a.bs_kv + (int64_t)pos * a.ld_kv + (int64_t)(hd - a.Hq) * a.d + j;
                    *(uint32_t*)dst = pack2bf(y1[0], y1[1]);
                    *(uint32_t*)(dst + a.d / 2) = pack2bf(y2[0], y2[1]);
                } else {
                    uint16_t* dst = a.vc + (int64_t)m * a.bs_kv + (int64_t)pos * a.ld_kv + (rows[0] - nqk);
                    *(u32x2*)dst = u32x2{pack2bf(v4[0], v4[1]), pack2bf(v4[2], v4[3])};
                }
            }
        }
    }
    }
}

// groups of 16 rows per wave (GR: the x fragments are re-used GR times) and waves per group block (KS: the K split) by the number of weight
// rows: as many rows per wave as still leaves >= ~1000 waves in flight.  A function of (weight rows, K) only, so a fused kernel and the
// launch sequence it replaces see the same sums.
void gemv_mfma_shape(int64_t units, int& gr, int& ks) {
    const int64_t groups = (units + 3) / 4;
    gr = 1;                                                  // measured (profiles/r5_gemv_rows.log): 2 / 4 groups per wave buy nothing on the wide shapes (gate|up 67 us
                                                             // either way: the x traffic was never the limit) and lose on the narrow ones (fewer waves)
    const int64_t blocks = (groups + gr - 1) / gr;
    ks = blocks <= 640 ? 4 : (blocks <= 1280 ? 2 : 1);
}

template <int MODE, int GR>
int launch_gemv_mfma_gr(GemvMfmaArgs& g, int64_t units, bool prenorm, hipStream_t s) {
    const int64_t blocks = ((units + 3) / 4 + GR - 1) / GR;
    const int bpw = (NT / 64) / g.ks;
    const unsigned grid = (unsigned)((blocks + bpw - 1) / bpw);
    if (prenorm) {
        if (g.f.M > 16) return MM355_EUNSUPPORTED;
        g.xs_stride = (((g.f.K + 31) >> 5) * 32 + 8) * 2;    // whole 32-column steps + 16 B per row: the 16 x rows of a fragment read fall on different banks
        const int lds = g.f.M * g.xs_stride;
        if (lds > 140 * 1024) return MM355_EUNSUPPORTED;
        static std::atomic<uint64_t> ok{0};                  // (one opt-in per device, to the largest size any call may ask for)
        if (mm_ensure_dynamic_lds((const void*)gemv_mfma_kernel<MODE, true, GR>, 140 * 1024, ok) != MM355_OK) return MM355_ELAUNCH;
        hipLaunchKernelGGL((gemv_mfma_kernel<MODE, true, GR>), dim3(grid), dim3(NT), lds, s, g);
    } else {
        if constexpr (GR == 1) {                             // weights coalesced + re-laid out through LDS, x rows through LDS windows
            // measured (profiles/r5_gemv_rows.log): the windows win wherever the waves of a workgroup share one K range (wide weights:
            // gate|up 47 -> 42 us at 8 rows, 53 -> 42 at 16, lm_head 208 -> 174) and, with K split over the waves, from nine rows on
            // (down 26.8 -> 23.6 at 16); below that the per-block barriers of four slices cost more than the L2 reads they replace
            if (g.f.M > 16) {                                // 17 .. 32 rows: two x row groups per workgroup, shapes with one K range per workgroup only
                if constexpr (MODE == 2) return MM355_EUNSUPPORTED;
                else {
                    if (g.ks != 1 || g.f.M > 32) return MM355_EUNSUPPORTED;
                    static std::atomic<uint64_t> ok2{0};
                    constexpr int XL2 = 2 * 1 * 16 * 2 * 544;
                    if (mm_ensure_dynamic_lds((const void*)gemv_mfma_kernel<MODE, false, 1, true, 1, 2>, XL2, ok2) != MM355_OK) return MM355_ELAUNCH;
                    hipLaunchKernelGGL((gemv_mfma_kernel<MODE, false, 1, true, 1, 2>), dim3(grid), dim3(NT), XL2, s, g);
                    return mm_launch_status();
                }
            }
            if (g.ks > 1 && g.f.M <= 8) { hipLaunchKernelGGL((gemv_mfma_kernel<MODE, false, 1, true>), dim3(grid), dim3(NT), 0, s, g); return mm_launch_status(); }
            const int xlds = 2 * g.ks * 16 * 544;
#define GX(KSC) do { static std::atomic<uint64_t> okx{0};                                                                                      \
                if (mm_ensure_dynamic_lds((const void*)gemv_mfma_kernel<MODE, false, 1, true, KSC>, 2 * 4 * 16 * 544, okx) != MM355_OK) return MM355_ELAUNCH; \
                hipLaunchKernelGGL((gemv_mfma_kernel<MODE, false, 1, true, KSC>), dim3(grid), dim3(NT), xlds, s, g); } while (0)
            if (g.ks == 4) GX(4); else if (g.ks == 2) GX(2); else GX(1);
#undef GX
        } else {
            hipLaunchKernelGGL((gemv_mfma_kernel<MODE, false, GR>), dim3(grid), dim3(NT), 0, s, g);
        }
    }
    return mm_launch_status();
}

// the MFMA kernel addresses weights and x rows with 32-bit buffer offsets, and marks a skipped prefetch by setting the top four bits
bool gemv_mfma_addressable(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw) {
    return (N - 1) * ldw * 2 + K * 2 < 0xf0000000ll && (M - 1) * ldx * 2 + K * 2 < 0xf0000000ll;
}

template <int MODE>
int launch_gemv_mfma(GemvMfmaArgs& g, int64_t units, bool prenorm, hipStream_t s) {
    int gr;
    gemv_mfma_shape(units, gr, g.ks);                        // (gr is 1: two / four groups of 16 rows per wave were measured and bought nothing; the kernel keeps the parameter)
    return launch_gemv_mfma_gr<MODE, 1>(g, units, prenorm, s);
}

// ------------------------------------------------------------------------------------------------ up to four rows: x in LDS, weights alone in the queue
// Round 4's kernels (gemv_kernel at the top, and a fused twin of it) read the x chunk of every trip from L2 INSIDE the loop, i.e. behind
// the weight loads they had just prefetched: the vector-memory counter retires in order, so each x chunk waited for the whole prefetch
// block and the pipeline never ran ahead -- 2.9 / 3.1 TB/s on the N = 4096 / 6144 projections (o: 11.8 us, q|k|v: 16.3 us), 5.2 on
// gate|up (45.5 us), 3.44 ms per token.  Here nothing but weight loads enters the queue once the stream runs:
//   * the x rows (and the norm weight) go FIRST, then the first NB trips of weights (a trip = 1024 k x 4 rows = 8 loads of 16 B per lane);
//     while those fly the workgroup reduces the rows (PRENORM: rmsnorm_fwd_kernel's order), forms bf16(w * bf16(x * rstd)) and parks the
//     rows in LDS (zero beyond K), whose reads have their own counter;
//   * the stream continues through a register ring: trip t is consumed, trip t + NB issued into its registers; behind the end the offset
//     carries the out-of-range mark (buffer loads return zeros without touching memory: no branch around a prefetch);
//   * chunk order, fma order and epilogues are those of gemv_kernel<MR, 1> (plain) and of the launch sequences the fused modes replace
//     (GEMV -> SwiGLU; GEMV -> RoPE + cache append): the same bits (a lane beyond K adds fma(x, 0, acc): nothing).
//   * rows longer than a window (16384 columns for one or two rows, 4096 for four) pass through LDS in WINDOWS (two buffers): the next
//     window's vectors are loaded while the current one is consumed and stored before the barrier that ends it -- any K;
//   * three and four rows (MR = 4) take v_dot2c_f32_bf16 on the packed pairs -- full rate (5.2 cycles per wave instruction, as
//     v_fmac_f32: tools/probes) and no unpacking: 16 M instead of 40 M vector instructions per chunk; fp32 accumulation either way.
//     (An eight-row instantiation was measured too -- 208 registers, two waves per SIMD, 8 x 16-byte LDS reads per chunk: gate|up 47 us --
//     and lost to the MFMA form above once that loaded its weights coalesced: five rows and more go there.)
// Measured (profiles/r5_decode_*): one row: o 8.3 us, q|k|v 12.3, gate|up 37.1 (6.3 TB/s), down 21.5 (was 22.5 with K split over four
// waves), lm_head 155 (was 170): 2.89 ms per token = 5.2 TB/s of weights; ring depth 1 .. 4 is within noise of it (the in-order x loads
// were the stall, not the depth).  Four rows: gate|up 41 us (5.7 TB/s).
template <int MR, int MODE, bool PRENORM, bool WINDOWS, int WT = 4>
__global__ __launch_bounds__(NT) void gemv_deep_kernel(GemvMfmaArgs g) {
    const GemvFusedArgs& a = g.f;
    constexpr int R = 4;
    constexpr bool DOT2 = MR > 2;                            // (the fma chain at four / eight rows: gate|up 48.8 / 89.7 us against 42.0 / 63.9)
    constexpr int NB = 2;                                    // trips in the register ring (measured at one row, 1 .. 4: 2.87 / 2.89 / 3.08 / 3.00 ms per token)
    constexpr int WK = WT * 1024;                            // an x window: WT trips = 4096 columns; 16384 for the plain GEMV of long rows
                                                             // (K = 14336 is then ONE window, no barrier in the stream: down 22.5 -> 21.7 us at one row, 25.1 -> 23.3 at four)
    constexpr int XV = WK / 8 / NT;                          // 16-byte vectors per thread, row and window (2; 8)
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];          // [1 or 2 windows][MR][wk] bf16 x rows as the dot products take them
    __shared__ float red[MR][NT / 64];
    __shared__ float rstd_s[MR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.x * (NT / 64) + wave;          // four weight rows
    const int K = a.K, M = a.M;
    const int ntrip = (K + 1023) >> 10, nwin = (ntrip + WT - 1) / WT, nv = K >> 3;
    const int wk = min(ntrip, WT) << 10;                     // elements per window row in LDS
    int rows[R];
    unit_rows<MODE>(a, unit, rows);
    bool live;
    if constexpr (MODE == 0) live = rows[0] < a.N;
    else if constexpr (MODE == 1) live = rows[1] < a.I;
    else live = rows[3] < a.N;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (uint32_t)((uint64_t)(a.N - 1) * a.ldw * 2 + (uint64_t)K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (uint32_t)((uint64_t)(M - 1) * a.ldx * 2 + (uint64_t)K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void*)(PRENORM ? a.norm_w : a.x), 0, (uint32_t)K * 2u, 0x00020000);
    const uint32_t OOB = 0xf0000000u;
    uint32_t wo[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wo[r] = (uint32_t)min(rows[r], a.N - 1) * (uint32_t)a.ldw * 2u;
    // this thread's vectors v = t + 256 i of window w of the x rows (and of the norm weight); beyond K: zeros, no memory access
    constexpr int MP = MR;
    u32x4 xr[MR][XV], nr[XV];
    auto stage_load = [&](int w, int m0, int m1) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = w * (WK / 8) + threadIdx.x + NT * i;
            const uint32_t sk = v < nv ? 0u : OOB;
#pragma unroll
            for (int m = 0; m < MR; ++m)
                if (m >= m0 && m < m1)
                    xr[m % (m1 - m0 == MR ? MR : MP)][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ((uint32_t)min(m, M - 1) * (uint32_t)a.ldx * 2u + (uint32_t)v * 16u) | sk, 0, 0);
            if constexpr (PRENORM) nr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsN, ((uint32_t)v * 16u) | sk, 0, 0);
        }
    };
    auto stage_store = [&](int w, int m0, int m1) {          // -> LDS window buffer w & 1, in the form the dot products take
        unsigned char* dst = xs + (size_t)(w & 1) * MR * wk * 2;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = threadIdx.x + NT * i;
            if (v * 8 < wk) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if (m < m0 || m >= m1) continue;
                    const int ms = m % (m1 - m0 == MR ? MR : MP);
                    u32x4 out = xr[ms][i];
                    if constexpr (PRENORM) {
                        float xv[8], nw[8];
                        unpack8(xr[ms][i], xv);
                        unpack8(nr[i], nw);
                        const float rs = rstd_s[m];
#pragma unroll
                        for (int e = 0; e < 8; ++e) xv[e] = nw[e] * round_bf(xv[e] * rs);
                        out = pack8(xv);
                    }
                    *(u32x4*)(dst + ((size_t)m * wk + v * 8) * 2) = out;
                }
            }
        }
    };
    auto issue = [&](u32x4 (&w)[2][R], int t) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int k = (t * 2 + ch) * 512 + lane * 8;
            const uint32_t sk = (live && k < K) ? 0u : OOB;
#pragma unroll
            for (int r = 0; r < R; ++r) w[ch][r] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (wo[r] + (uint32_t)k * 2u) | sk, 0, 2);
        }
    };
    // rmsnorm_fwd_kernel's reduction (thread t sums elements 8 (t + 256 i) .. + 7 in order, block_sum<256>), all rows in one pass
    auto finish_norm = [&](float (&ss)[MR]) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float w = wave_sum(ss[m]);
            if (lane == 0) red[m][wave] = w;
        }
        __syncthreads();
        if (threadIdx.x < MR) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NT / 64; ++i) t += red[threadIdx.x][i];
            rstd_s[threadIdx.x] = rsqrtf(t / (float)K + a.eps);
        }
        __syncthreads();
    };
    if constexpr (PRENORM) {
        if (nwin > 1) {                                      // rows longer than a window: their sums of squares first (x from L2, read again below)
            float ss[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) ss[m] = 0.f;
            for (int v = threadIdx.x; v < nv; v += NT) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    float xv[8];
                    unpack8(*(const u32x4*)(a.x + (int64_t)min(m, M - 1) * a.ldx + v * 8), xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss[m] += xv[e] * xv[e];
                }
            }
            finish_norm(ss);
        }
    }
    // ---- (1) the first x window goes FIRST into the vector-memory queue, (2) then the first NB trips of the weight stream
    stage_load(0, 0, MR);
    u32x4 wb[NB][2][R];
#pragma unroll
    for (int j = 0; j < NB; ++j) issue(wb[j], j);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (3) the rows -> LDS
    if constexpr (PRENORM) {
        if (nwin == 1) {
            float ss[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                ss[m] = 0.f;
#pragma unroll
                for (int i = 0; i < XV; ++i) {
                    float xv[8];
                    unpack8(xr[m][i], xv);                   // (beyond K: zeros)
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss[m] += xv[e] * xv[e];
                }
            }
            finish_norm(ss);
        }
    }
    stage_store(0, 0, MR);
    __syncthreads();
    // ---- (4) the stream
    float acc[MR][R];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
    auto consume = [&](const u32x4 (&w)[2][R], int win, int j) {          // trip j of window win
        const unsigned char* src = xs + (size_t)(win & 1) * MR * wk * 2;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int k = min((j * 2 + ch) * 512, wk - 512) + lane * 8;       // (a trip behind the end is all zeros: any x will do)
            if constexpr (!DOT2) {                           // gemv_kernel's arithmetic: fp32 fma chain over the eight elements
                float xf[MR][8];
#pragma unroll
                for (int m = 0; m < MR; ++m) unpack8(*(const u32x4*)(src + ((size_t)m * wk + k) * 2), xf[m]);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float wf[8];
                    unpack8(w[ch][r], wf);
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[m][r] = fmaf(xf[m][e], wf[e], acc[m][r]);
                }
            } else {                                         // 3 .. 4 rows: v_dot2c_f32_bf16 on the packed pairs (no unpacking: 16 MR VALU ops per chunk, not 40 MR)
                u32x4 xv[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m) xv[m] = *(const u32x4*)(src + ((size_t)m * wk + k) * 2);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t we = w[ch][r][e], xe = xv[m][e];   // (scalars first: a bit_cast OF a vector element reads element 0 whatever e -- hipcc 7.2)
                            acc[m][r] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(mm_bf16x2, we), __builtin_bit_cast(mm_bf16x2, xe), acc[m][r], false);
                        }
            }
        }
    };
    static_assert(WT % NB == 0, "the ring position of a trip must not depend on the window");
    for (int win = 0; win < nwin; ++win) {
        const int nt = min(WT, ntrip - win * WT);            // trips of this window
        if constexpr (WINDOWS) {                             // (a row that fits one window: no staging registers live across the stream)
            stage_load(win + 1, 0, MR);                      // the NEXT window (behind the last: out of range, no access) lands while this one is consumed
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int j0 = 0; j0 < nt; j0 += NB) {
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) {
                consume(wb[jj], win, j0 + jj);               // (a trip behind the end is all zeros)
                // (the sums are pure arithmetic: without this pin hipcc sinks them below the loads that follow, renaming the ring into 256
                // registers; the empty statement keeps "consume trip t, then refill its registers")
#pragma unroll
                for (int m = 0; m < MR; ++m) asm volatile("" : "+v"(acc[m][0]), "+v"(acc[m][1]), "+v"(acc[m][2]), "+v"(acc[m][3]) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
                issue(wb[jj], win * WT + j0 + jj + NB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (WINDOWS) {                             // into the OTHER buffer: everyone left it at the barrier before this window
            stage_store(win + 1, 0, MR);
            __syncthreads();
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = wave_sum(acc[m][r]);
    if (!live) return;
    if constexpr (MODE == 0) {                               // gemv_kernel's epilogue: lane (m * R + r) finishes output (m, rows[r])
        if (lane < MR * R) {
            const int m = lane / R, r = lane % R, n = rows[0] + r;
            if (m < M && n < a.N) {
                float v = 0.f;
#pragma unroll
                for (int mm = 0; mm < MR; ++mm)
#pragma unroll
                    for (int rr = 0; rr < R; ++rr)
                        if (mm == m && rr == r) v = acc[mm][rr];
                const uint32_t flags = g.flags;
                if (flags & MM355_GEMM_BIAS) v += bf2f(g.bias[n]);
                if (flags & MM355_GEMM_GELU_ERF) v = gelu_erf_f(v);
                if (flags & MM355_GEMM_GELU_TANH) v = gelu_tanh_f(v);
                if (flags & MM355_GEMM_RESIDUAL) v += bf2f(g.res[(int64_t)m * g.ldr + n]);
                if (flags & MM355_GEMM_OUT_F32) ((float*)g.y)[(int64_t)m * g.ldy + n] = v;
                else ((uint16_t*)g.y)[(int64_t)m * g.ldy + n] = f2bf(v);
            }
        }
        return;
    } else {
        if (lane >= M) return;                               // the fused epilogues: lane m finishes the unit's outputs of row m
        float v4[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = 0.f;
#pragma unroll
            for (int mm = 0; mm < MR; ++mm)
                if (mm == lane) t = acc[mm][r];
            v4[r] = round_bf(t);                             // what the unfused GEMV stores
        }
        const int m = lane;
        if constexpr (MODE == 1) {
            float o[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) o[e] = round_bf(v4[e] / (1.0f + __expf(-v4[e]))) * v4[2 + e];
            *(uint32_t*)(a.out + (int64_t)m * a.ld_out + rows[0]) = pack2bf(o[0], o[1]);
        } else {
            const int nqk = (a.Hq + a.Hkv) * a.d;
            const int pos = a.positions[m];
            if (rows[0] < nqk) {
                const int hd = rows[0] / a.d, j = rows[0] % a.d;
                float y1[2], y2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float c = bf2f(a.cos_t[(int64_t)pos * a.d + j + e]), sn = bf2f(a.sin_t[(int64_t)pos * a.d + j + e]);
                    y1[e] = round_bf(v4[e] * c) + round_bf(-v4[2 + e] * sn);
                    y2[e] = round_bf(v4[2 + e] * c) + round_bf(v4[e] * sn);
                }
                uint16_t* dst = hd < a.Hq ? a.out + (int64_t)m * a.ld_out + rows[0]
                                          : a.kc + (int64_t)m * a.bs_kv + (int64_t)pos * a.ld_kv + (int64_t)(hd - a.Hq) * a.d + j;
                *(uint32_t*)dst = pack2bf(y1[0], y1[1]);
                *(uint32_t*)(dst + a.d / 2) = pack2bf(y2[0], y2[1]);
            } else {
                uint16_t* dst = a.vc + (int64_t)m * a.bs_kv + (int64_t)pos * a.ld_kv + (rows[0] - nqk);
                *(u32x2*)dst = u32x2{pack2bf(v4[0], v4[1]), pack2bf(v4[2], v4[3])};
            }
        }
    }
}

bool gemv_deep_applies(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw) {
    return M <= 4 && gemv_mfma_addressable(M, N, K, ldx, ldw);
}

template <int MODE>
int launch_gemv_deep(GemvMfmaArgs& g, int64_t units, bool prenorm, hipStream_t s) {
    const unsigned grid = (unsigned)((units + NT / 64 - 1) / (NT / 64));
    const int ntrip = (g.f.K + 1023) >> 10;
    const int mr = g.f.M == 1 ? 1 : (g.f.M == 2 ? 2 : 4);
    const int wt = (MODE == 0 && !prenorm && ntrip > 4 && ntrip <= 16) ? 16 : 4;   // trips per x window (gemv_deep_kernel: WT)
    const int lds = (ntrip > wt ? 2 : 1) * mr * (ntrip < wt ? ntrip : wt) * 1024 * 2;    // <= 64 KiB unless two rows are longer than 16384 columns
#define GD4(MR, PN, WN) do { if (lds > 65536) { static std::atomic<uint64_t> ok{0};                                                            \
            if (mm_ensure_dynamic_lds((const void*)gemv_deep_kernel<MR, MODE, PN, WN>, 128 * 1024, ok) != MM355_OK) return MM355_ELAUNCH; }     \
        hipLaunchKernelGGL((gemv_deep_kernel<MR, MODE, PN, WN>), dim3(grid), dim3(NT), lds, s, g); } while (0)
#define GD3(MR, PN) do { if (wt == 16) { if constexpr (MODE == 0 && !PN) { static std::atomic<uint64_t> ok16{0};                               \
            if (mm_ensure_dynamic_lds((const void*)gemv_deep_kernel<MR, 0, false, false, 16>, 128 * 1024, ok16) != MM355_OK) return MM355_ELAUNCH; \
            hipLaunchKernelGGL((gemv_deep_kernel<MR, 0, false, false, 16>), dim3(grid), dim3(NT), lds, s, g); } }                               \
        else if (ntrip > wt) GD4(MR, PN, true); else GD4(MR, PN, false); } while (0)
#define GD(MR) do { if (prenorm) { if constexpr (MODE != 0) GD3(MR, true); } else GD3(MR, false); } while (0)
    if (mr == 1) GD(1);
    else if (mr == 2) GD(2);
    else GD(4);
#undef GD
#undef GD3
#undef GD4
    return mm_launch_status();
}

// ------------------------------------------------------------------------------------------------ RoPE + cache append
// One new qkv row per sample: rotate q and k at position positions[b] (HF rounding order, as rope_qk_kernel), leave q in
// place and write the rotated k and the v row into cache row positions[b].  Positions come from DEVICE memory so that the
// whole per-token step is replayable as a hipGraph.
__global__ __launch_bounds__(NT) void rope_kv_append_kernel(uint16_t* __restrict__ qkv, int64_t ld, int Hq, int Hkv, int d,
                                                            const uint16_t* __restrict__ cos_t, const uint16_t* __restrict__ sin_t,
                                                            const int32_t* __restrict__ positions, uint16_t* __restrict__ kc,
                                                            uint16_t* __restrict__ vc, int64_t ld_kv, int64_t bs_kv) {
    const int b = blockIdx.y;
    const int pos = positions[b];
    const int half = d >> 1, vph = half >> 3;
    const int H = Hq + Hkv;
    uint16_t* row = qkv + (int64_t)b * ld;
    uint16_t* krow = kc + (int64_t)b * bs_kv + (int64_t)pos * ld_kv;
    uint16_t* vrow = vc + (int64_t)b * bs_kv + (int64_t)pos * ld_kv;
    const int rot = H * vph, cpy = Hkv * d / 8;
    for (int i = blockIdx.x * NT + threadIdx.x; i < rot + cpy; i += gridDim.x * NT) {
        if (i < rot) {
            const int v = i % vph, hd = i / vph;
            uint16_t* p1 = row + (int64_t)hd * d + v * 8;
            uint16_t* p2 = p1 + half;
            float x1[8], x2[8], c[8], sn[8], y1[8], y2[8];
            unpack8(*(const u32x4*)p1, x1);
            unpack8(*(const u32x4*)p2, x2);
            unpack8(*(const u32x4*)(cos_t + (int64_t)pos * d + v * 8), c);
            unpack8(*(const u32x4*)(sin_t + (int64_t)pos * d + v * 8), sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                y1[e] = round_bf(x1[e] * c[e]) + round_bf(-x2[e] * sn[e]);
                y2[e] = round_bf(x2[e] * c[e]) + round_bf(x1[e] * sn[e]);
            }
            const u32x4 o1 = pack8(y1), o2 = pack8(y2);
            if (hd < Hq) { *(u32x4*)p1 = o1; *(u32x4*)p2 = o2; }
            else {
                uint16_t* kd = krow + (int64_t)(hd - Hq) * d + v * 8;
                *(u32x4*)kd = o1; *(u32x4*)(kd + half) = o2;
            }
        } else {
            const int j = i - rot;
            *(u32x4*)(vrow + j * 8) = *(const u32x4*)(row + (int64_t)H * d + j * 8);
        }
    }
}

// ------------------------------------------------------------------------------------------------ attention, decode shape
constexpr int CH = 256;                                      // keys per workgroup (one per thread)
constexpr int GMAX = 8;                                      // query heads per KV head handled by one workgroup

// partial record per (b, q head, split): [m, l, o[d]] floats
template <int G>
__global__ __launch_bounds__(NT) void attn_decode_split_kernel(const uint16_t* __restrict__ q, int64_t ld_q, const uint16_t* __restrict__ kc,
                                                               const uint16_t* __restrict__ vc, int64_t ld_kv, int64_t bs_kv,
                                                               const int32_t* __restrict__ kv_lens, float* __restrict__ ws, int nsplit,
                                                               int Hq, int Hkv, int d, float scale, uint16_t* __restrict__ o, int64_t ld_o,
                                                               int* __restrict__ counters) {
    __shared__ float sq[G][128];                             // query rows (fp32, pre-scaled)
    __shared__ float sp[G][CH];                              // probabilities of this chunk
    __shared__ float red[G][NT / 64];
    __shared__ float so[NT / 64][G][128];                    // per-wave partial outputs
    const int s = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kv_len = kv_lens[b];
    const int rec = d + 2;
    float* wrec = ws + (((int64_t)b * Hq + hk * G) * nsplit + s) * rec;      // record of q head hk*G + g: + g * nsplit * rec
    const int k0 = s * CH;
    __shared__ int last_s;
    if (k0 >= kv_len) {                                      // chunk beyond the cache: empty partial
        if (tid < G) { float* w = wrec + (int64_t)tid * nsplit * rec; w[0] = -INFINITY; w[1] = 0.f; }
    } else {
    for (int i = tid; i < G * d; i += NT) {
        const int g = i / d, c = i % d;
        sq[g][c] = bf2f(q[(int64_t)b * ld_q + (int64_t)(hk * G + g) * d + c]) * scale;
    }
    __syncthreads();
    // ---- scores: lane = (key of a group of four, 16-B chunk of d); a wave walks its 64 keys four at a time, every K row is one
    //      coalesced 256-B read, the query chunks sit in registers, the 16 chunk partials are folded with DPP row shifts
    {
        const int dc = lane & 15, kq = lane >> 4;
        float qreg[G][8];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) qreg[g][e] = (dc * 8 < d) ? sq[g][dc * 8 + e] : 0.f;
        // all sixteen K rows of this lane are requested before the first one is used (independent 16-B loads in flight)
        u32x4 kraw[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = k0 + wave * 64 + i * 4 + kq;
            kraw[i] = (kk < kv_len && dc * 8 < d) ? *(const u32x4*)(kc + (int64_t)b * bs_kv + (int64_t)kk * ld_kv + (int64_t)hk * d + dc * 8)
                                                  : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kl = wave * 64 + i * 4 + kq;               // key inside the chunk
            const int kk = k0 + kl;
            float kf[8];
            unpack8(kraw[i], kf);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float v = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) v = fmaf(kf[e], qreg[g][e], v);
                // sum over the 16 lanes of a row: lane 15 ends up with the total
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
                if (dc == 15) sp[g][kl] = (kk < kv_len) ? v : -INFINITY;
            }
        }
    }
    __syncthreads();
    const int key = k0 + tid;
    float sc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) sc[g] = sp[g][tid];
    __syncthreads();                                         // sp is rewritten with the probabilities below
    float mx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float w = wave_max(sc[g]);
        if (lane == 0) red[g][wave] = w;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) mx[g] = fmaxf(fmaxf(red[g][0], red[g][1]), fmaxf(red[g][2], red[g][3]));
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float p = (key < kv_len) ? __expf(sc[g] - mx[g]) : 0.f;
        sp[g][tid] = p;
        const float w = wave_sum(p);
        if (lane == 0) red[g][wave] = w;
    }
    __syncthreads();
    // ---- o = sum_key p[key] V[key]: thread = (16-B chunk of d, key group)
    const int dc = tid & 15, kg = tid >> 4;                  // 16 chunks x 16 key groups
    float acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    {
        const int kend = min(CH, kv_len - k0);
        u32x4 vraw[16];                                      // sixteen independent 16-B loads in flight
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kg + i * 16;
            vraw[i] = (kk < kend && dc * 8 < d) ? *(const u32x4*)(vc + (int64_t)b * bs_kv + (int64_t)(k0 + kk) * ld_kv + (int64_t)hk * d + dc * 8)
                                                : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kg + i * 16;
            float vf[8];
            unpack8(vraw[i], vf);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float p = (kk < kend) ? sp[g][kk] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e]);
            }
        }
    }
    // the four key groups of a wave are lanes l, l^16, l^32, l^48: fold them, one row of partials per wave
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[g][e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16 && dc * 8 < d) so[wave][g][dc * 8 + e] = v;
        }
    __syncthreads();
    for (int i = tid; i < G * d; i += NT) {
        const int g = i / d, c = i % d;
        wrec[(int64_t)g * nsplit * rec + 2 + c] = (so[0][g][c] + so[1][g][c]) + (so[2][g][c] + so[3][g][c]);
    }
    if (tid < G) {
        float* w = wrec + (int64_t)tid * nsplit * rec;
        w[0] = mx[tid];
        w[1] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
    }
    }
    // ---- the chunk that arrives last merges the partials of its G query heads (round 4: no second launch).  Device-scope fences on
    //      both sides of the counter: the other chunks ran on other XCDs (their records sit in other L2s until written back)
    if (counters == nullptr) return;
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const int prev = atomicAdd(&counters[b * Hkv + hk], 1);
        last_s = prev == nsplit - 1;
        if (last_s) counters[b * Hkv + hk] = 0;              // ready for the next launch (graph replay)
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    for (int i = tid; i < G * d; i += NT) {                   // attn_decode_merge_kernel's arithmetic and order
        const int g = i / d, c = i % d;
        const float* w = ws + (((int64_t)b * Hq + hk * G + g) * nsplit) * rec;
        float Mx = -INFINITY;
        for (int t = 0; t < nsplit; ++t) Mx = fmaxf(Mx, __builtin_nontemporal_load(w + (int64_t)t * rec));
        float l = 0.f, acc1 = 0.f;
        for (int t = 0; t < nsplit; ++t) {
            const float mm = __builtin_nontemporal_load(w + (int64_t)t * rec);
            if (mm == -INFINITY) continue;
            const float f = __expf(mm - Mx);
            l += f * __builtin_nontemporal_load(w + (int64_t)t * rec + 1);
            acc1 += f * __builtin_nontemporal_load(w + (int64_t)t * rec + 2 + c);
        }
        o[(int64_t)b * ld_o + (int64_t)(hk * G + g) * d + c] = f2bf(l > 0.f ? acc1 / l : 0.f);
    }
}

// One workgroup of 1024 threads per (sample, KV head, 1024-key group): four sub-blocks of 256 threads take one 256-key chunk each (the
// arithmetic of attn_decode_split_kernel), their partials meet in LDS, and while the cache holds <= 1024 rows the lone active workgroup
// writes the output row itself -- no workspace round trip, no device-scope fence, no counter (round 4: 23 -> ~10 us per layer at a
// 600-row cache).  Longer caches: one record per 1024-key group, merged by the group that arrives last (counters as above).
template <int G>
__global__ __launch_bounds__(1024) void attn_decode_wide_kernel(const uint16_t* __restrict__ q, int64_t ld_q, const uint16_t* __restrict__ kc,
                                                                const uint16_t* __restrict__ vc, int64_t ld_kv, int64_t bs_kv,
                                                                const int32_t* __restrict__ kv_lens, float* __restrict__ ws, int ngroup,
                                                                int Hq, int Hkv, int d, float scale, uint16_t* __restrict__ o, int64_t ld_o,
                                                                int* __restrict__ counters, int kvdiv) {
    constexpr int SB = 4;
    constexpr int NB = G >= 8 ? 2 : (G >= 4 ? 4 : 8);                       // K / V rows per lane in flight (128 registers per thread at 16 waves per CU)
    constexpr int NBK = G == 1 ? 16 : NB;                                       // one query head: all sixteen K rows of a lane at once,
    constexpr int VPRE = G == 1 ? 16 : 0;                                       //   and its V rows issued before the softmax barriers (G = 2 spills with either)
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float (*sq)[128] = (float (*)[128])lds_f;                                   // [G][128] (rounds 4-5: query rows; unused since q goes straight into registers)
    float (*sp)[G][CH] = (float (*)[G][CH])(lds_f + G * 128);                   // [SB][G][CH] scores, then probabilities
    float (*redm)[G][4] = (float (*)[G][4])(lds_f + G * 128 + SB * G * CH);     // [SB][G][4] per-wave maxima
    float (*reds)[G][4] = (float (*)[G][4])(lds_f + G * 128 + SB * G * CH + SB * G * 4);
    float (*so)[G][128] = (float (*)[G][128])(lds_f + G * 128 + SB * G * CH + 2 * SB * G * 4);   // [SB * 4 waves][G][128]
    __shared__ int last_s;
    const int grp = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, sb = tid >> 8, t = tid & 255, lane = tid & 63, wave = t >> 6;
    const int kv_len = kv_lens[b];
#if defined(MM355_AD_STOP) && MM355_AD_STOP == 0                                 // TIMING-ONLY builds (tools/build_ad_stop.sh): the kernel cut short after a phase
    if (kv_len >= 0) return;
#endif
    const int ngr_live = (kv_len + SB * CH - 1) / (SB * CH);                    // groups that hold keys
    const int rec = d + 2;
    float* wrec = ws + (((int64_t)b * Hq + hk * G) * ngroup + grp) * rec;
    const int k0 = (grp * SB + sb) * CH;
    const bool active = k0 < kv_len;                                            // this sub-block's chunk holds keys
    // Round 6: where the time of this kernel goes.  At one sequence and a 512-row cache it moved 2 MB in 10.5 us, and the obvious reading -- a
    // chain of trips to HBM: q through LDS, two batches of K rows, three barriers of softmax, two batches of V rows -- was WRONG: with every
    // load of a thread issued up front (q straight into registers, sixteen K rows, the V rows before the softmax barriers) it took 11-14 us.
    // The workgroup is bound by INSTRUCTION ISSUE: sixteen waves on one CU = four per SIMD, each running the whole stream, and
    //   * the sub-blocks whose 256-key chunk lies beyond the cache ran it too, every lane masked off (half the waves at 512 rows): they now
    //     skip the score, softmax and PV phases (wave-uniform branches; only the barriers are shared) -- their partials were never read;
    //   * a load under a LANE predicate is a branch, and hipcc sank the whole address computation, an integer division by kvdiv included,
    //     into each of the 32 branches (~50 instructions per load): rows are now loaded unconditionally from clamped addresses (a row beyond
    //     the cache reads the last cached row, a chunk of d beyond the head reads chunk 0 -- finite values that meet a score of -inf, a
    //     probability of 0 or a store nobody makes) from two base pointers computed once;
    //   * q goes straight into the lanes' registers (no LDS copy, one barrier less).
    // Same products and the same order of additions per output as before: bit-identical results.
    const int dc8_t = ((t & 15) * 8 < d) ? (t & 15) * 8 : 0;                   // (lane & 15 == t & 15: both phases address the same chunk)
    const uint16_t* const kbase = kc + (int64_t)b * bs_kv + (int64_t)(hk / kvdiv) * d + dc8_t;
    const uint16_t* const vbase = vc + (int64_t)b * bs_kv + (int64_t)(hk / kvdiv) * d + dc8_t;
    (void)sq;
    if (grp * SB * CH < kv_len) {
        // ---- scores (batches of NB K rows per lane: 128 registers per thread at 16 waves per CU)
        if (active) {
            const int dc = lane & 15, kq = lane >> 4;
            float qreg[G][8];
            {
                // one 16-B load where q allows it, else eight 2-byte ones (q carries no alignment contract)
                const bool qvec = ((((uintptr_t)q) | ((uintptr_t)ld_q * 2u)) & 15u) == 0;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint16_t* qp = q + (int64_t)b * ld_q + (int64_t)(hk * G + g) * d + dc8_t;
                    u32x4 qh;
                    if (qvec) {
                        qh = *(const u32x4*)qp;
                    } else {
                        uint32_t w[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)qp[2 * e] | ((uint32_t)qp[2 * e + 1] << 16);
                        qh = u32x4{w[0], w[1], w[2], w[3]};
                    }
                    float qf[8];
                    unpack8(qh, qf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) qreg[g][e] = (dc * 8 < d) ? qf[e] * scale : 0.f;
                }
            }
#pragma unroll 1
            for (int h = 0; h < 16 / NBK; ++h) {
                u32x4 kraw[NBK];
#pragma unroll
                for (int i = 0; i < NBK; ++i) {
                    const int kk = k0 + wave * 64 + (h * NBK + i) * 4 + kq;
                    kraw[i] = *(const u32x4*)(kbase + (int64_t)min(kk, kv_len - 1) * ld_kv);
                }
#pragma unroll
                for (int i = 0; i < NBK; ++i) {
                    const int kl = wave * 64 + (h * NBK + i) * 4 + kq;
                    const int kk = k0 + kl;
                    float kf[8];
                    unpack8(kraw[i], kf);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        float v = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v = fmaf(kf[e], qreg[g][e], v);
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
                        if (dc == 15) sp[sb][g][kl] = (kk < kv_len) ? v : -INFINITY;
                    }
                }
            }
        }
        u32x4 vpre[VPRE ? VPRE : 1];
        if constexpr (VPRE > 0) {                            // the first V rows of the PV phase (its thread mapping), in flight across the softmax barriers
            if (active) {
#pragma unroll
                for (int i = 0; i < VPRE; ++i) vpre[i] = *(const u32x4*)(vbase + (int64_t)min(k0 + (t >> 4) + i * 16, kv_len - 1) * ld_kv);
            }
        }
        __syncthreads();
#if defined(MM355_AD_STOP) && MM355_AD_STOP == 1
        if (kv_len >= 0) return;
#endif
        const int key = k0 + t;
        float sc[G];
        if (active) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                sc[g] = sp[sb][g][t];
                const float w = wave_max(sc[g]);
                if (lane == 0) redm[sb][g][wave] = w;
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float mxg = fmaxf(fmaxf(redm[sb][g][0], redm[sb][g][1]), fmaxf(redm[sb][g][2], redm[sb][g][3]));
                const float p = (key < kv_len) ? __expf(sc[g] - mxg) : 0.f;
                sp[sb][g][t] = p;                             // (own slot: read above by this thread only)
                const float w = wave_sum(p);
                if (lane == 0) reds[sb][g][wave] = w;
            }
        }
        __syncthreads();
#if defined(MM355_AD_STOP) && MM355_AD_STOP == 2
        if (kv_len >= 0) return;
#endif
        // ---- o = sum_key p[key] V[key]: thread = (16-B chunk of d, key group of 16), batches of NB V rows
        if (active) {
            const int dc = t & 15, kg = t >> 4;
            float acc[G][8];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
            const int kend = min(CH, kv_len - k0);
            auto pv = [&](const u32x4& vr, int kk) {
                float vf[8];
                unpack8(vr, vf);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float p = (kk < kend) ? sp[sb][g][kk] : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e]);
                }
            };
            if constexpr (VPRE > 0) {
#pragma unroll
                for (int i = 0; i < VPRE; ++i) pv(vpre[i], kg + i * 16);
            }
            constexpr int NBV = (16 - VPRE) < NB ? (16 - VPRE ? 16 - VPRE : 1) : NB;
#pragma unroll 1
            for (int r0 = VPRE; r0 < 16; r0 += NBV) {
                u32x4 vraw[NBV];
#pragma unroll
                for (int i = 0; i < NBV; ++i) vraw[i] = *(const u32x4*)(vbase + (int64_t)min(k0 + kg + (r0 + i) * 16, kv_len - 1) * ld_kv);
#pragma unroll
                for (int i = 0; i < NBV; ++i) pv(vraw[i], kg + (r0 + i) * 16);
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc[g][e];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (lane < 16 && dc * 8 < d) so[sb * 4 + wave][g][dc * 8 + e] = v;
                }
        }
        __syncthreads();
#if defined(MM355_AD_STOP) && MM355_AD_STOP == 3
        if (kv_len >= 0) return;
#endif
        // ---- the four chunks of this group -> one (m, l, o[d]) per query head, in a fixed order
        for (int i = tid; i < G * d; i += 1024) {
            const int g = i / d, c = i % d;
            float M4[SB], Mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const bool on = (grp * SB + u) * CH < kv_len;
                M4[u] = on ? fmaxf(fmaxf(redm[u][g][0], redm[u][g][1]), fmaxf(redm[u][g][2], redm[u][g][3])) : -INFINITY;
                Mx = fmaxf(Mx, M4[u]);
            }
            float l = 0.f, acc1 = 0.f;
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (M4[u] == -INFINITY) continue;
                const float f = __expf(M4[u] - Mx);
                l += f * ((reds[u][g][0] + reds[u][g][1]) + (reds[u][g][2] + reds[u][g][3]));
                acc1 += f * ((so[u * 4 + 0][g][c] + so[u * 4 + 1][g][c]) + (so[u * 4 + 2][g][c] + so[u * 4 + 3][g][c]));
            }
            if (ngr_live == 1) {                              // the whole cache was this group's: done
                o[(int64_t)b * ld_o + (int64_t)(hk * G + g) * d + c] = f2bf(l > 0.f ? acc1 / l : 0.f);
            } else {
                float* w = wrec + (int64_t)g * ngroup * rec;
                w[2 + c] = acc1;
                if (c == 0) { w[0] = Mx; w[1] = l; }
            }
        }
    } else if (ngr_live > 1 && tid < G) {                     // group beyond the cache: empty record
        float* w = wrec + (int64_t)tid * ngroup * rec;
        w[0] = -INFINITY; w[1] = 0.f;
    }
    (void)active;
    if (ngr_live == 0 && grp == 0) {                          // an empty cache (kv_len == 0): a zero row, as the split kernel writes it
        for (int i = tid; i < G * d; i += 1024) o[(int64_t)b * ld_o + (int64_t)(hk * G + i / d) * d + i % d] = 0;
    }
    if (ngr_live <= 1) return;                                // (uniform over the grid: kv_len is per sample, read by every group of it)
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const int prev = atomicAdd(&counters[b * Hkv + hk], 1);
        last_s = prev == ngroup - 1;
        if (last_s) counters[b * Hkv + hk] = 0;
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    for (int i = tid; i < G * d; i += 1024) {
        const int g = i / d, c = i % d;
        const float* w = ws + (((int64_t)b * Hq + hk * G + g) * ngroup) * rec;
        float Mx = -INFINITY;
        for (int u = 0; u < ngroup; ++u) Mx = fmaxf(Mx, __builtin_nontemporal_load(w + (int64_t)u * rec));
        float l = 0.f, acc1 = 0.f;
        for (int u = 0; u < ngroup; ++u) {
            const float mm = __builtin_nontemporal_load(w + (int64_t)u * rec);
            if (mm == -INFINITY) continue;
            const float f = __expf(mm - Mx);
            l += f * __builtin_nontemporal_load(w + (int64_t)u * rec + 1);
            acc1 += f * __builtin_nontemporal_load(w + (int64_t)u * rec + 2 + c);
        }
        o[(int64_t)b * ld_o + (int64_t)(hk * G + g) * d + c] = f2bf(l > 0.f ? acc1 / l : 0.f);
    }
}

// G = query heads per workgroup; Hgrp = Hq / G workgroups per sample and key group, each reading cache head (its index) / kvdiv
template <int G>
int launch_wide(const uint16_t* q, int64_t ld_q, const uint16_t* kc, const uint16_t* vc, int64_t ld_kv, int64_t bs_kv, const int32_t* kv_lens,
                float* ws, int ngroup, int B, int Hq, int Hgrp, int d, float scale, uint16_t* o, int64_t ld_o, int* counters, hipStream_t s, int kvdiv = 1) {
    constexpr int LDS = (G * 128 + 4 * G * CH + 2 * 4 * G * 4 + 16 * G * 128) * 4;
    static std::atomic<uint64_t> lds_ok{0};
    if (LDS > 65536 && mm_ensure_dynamic_lds((const void*)attn_decode_wide_kernel<G>, LDS, lds_ok) != MM355_OK) return MM355_ELAUNCH;
    hipLaunchKernelGGL(attn_decode_wide_kernel<G>, dim3((unsigned)ngroup, (unsigned)Hgrp, (unsigned)B), dim3(1024), LDS, s, q, ld_q, kc, vc, ld_kv,
                       bs_kv, kv_lens, ws, ngroup, Hq, Hgrp, d, scale, o, ld_o, counters, kvdiv);
    return mm_launch_status();
}

template <int G>
int launch_split(const uint16_t* q, int64_t ld_q, const uint16_t* kc, const uint16_t* vc, int64_t ld_kv, int64_t bs_kv, const int32_t* kv_lens,
                 float* ws, int nsplit, int B, int Hq, int Hkv, int d, float scale, uint16_t* o, int64_t ld_o, int* counters, hipStream_t s) {
    hipLaunchKernelGGL(attn_decode_split_kernel<G>, dim3((unsigned)nsplit, (unsigned)Hkv, (unsigned)B), dim3(NT), 0, s, q, ld_q, kc, vc, ld_kv,
                       bs_kv, kv_lens, ws, nsplit, Hq, Hkv, d, scale, o, ld_o, counters);
    return mm_launch_status();
}

}  // namespace

extern "C" int mm355_gemv_bf16(const mm355_bf16* x, int64_t ldx, const mm355_bf16* W, int64_t ldw, void* y, int64_t ldy, int64_t M, int64_t N,
                               int64_t K, const mm355_bf16* bias, const mm355_bf16* residual, int64_t ldr, uint32_t flags, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0) return MM355_EINVAL;
    if (M > 32) return MM355_EUNSUPPORTED;                   // more rows: mm355_gemm_bf16 (17 .. 32: wide weights only, see launch_gemv_mfma_gr)
    if ((K & 7) || (ldx & 7) || (ldw & 7) || !mm_aligned16(x) || !mm_aligned16(W)) return MM355_EINVAL;
    if ((flags & MM355_GEMM_BIAS) && !bias) return MM355_EINVAL;
    if ((flags & MM355_GEMM_RESIDUAL) && !residual) return MM355_EINVAL;
    if ((flags & MM355_GEMM_GELU_ERF) && (flags & MM355_GEMM_GELU_TANH)) return MM355_EINVAL;
    if (flags & MM355_GEMM_ACCUMULATE) return MM355_EUNSUPPORTED;
    if (N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (M > 4) {                                             // 5 .. 16 rows: the MFMA form (up to four rows: the VALU kernels below)
        if (!gemv_mfma_addressable(M, N, K, ldx, ldw)) return MM355_EUNSUPPORTED;
        GemvMfmaArgs g = {};
        g.f.x = x; g.f.ldx = ldx; g.f.W = W; g.f.ldw = ldw; g.f.M = (int)M; g.f.N = (int)N; g.f.K = (int)K;
        g.y = y; g.ldy = ldy; g.bias = bias; g.res = residual; g.ldr = ldr; g.flags = flags;
        return launch_gemv_mfma<0>(g, (N + 3) / 4, false, s);
    }
    if (gemv_deep_applies(M, N, K, ldx, ldw)) {
        GemvMfmaArgs g = {};
        g.f.x = x; g.f.ldx = ldx; g.f.W = W; g.f.ldw = ldw; g.f.M = (int)M; g.f.N = (int)N; g.f.K = (int)K;
        g.y = y; g.ldy = ldy; g.bias = bias; g.res = residual; g.ldr = ldr; g.flags = flags;
        return launch_gemv_deep<0>(g, (N + 3) / 4, false, s);
    }
    // a weight beyond 32-bit byte offsets: the plain stream (x re-read from L2 per chunk); few rows of a long K split K over the waves
    // of a workgroup
    const int ksplit = (N <= 8192 && K >= 8192) ? 4 : 1;
    const int rows_per_wg = 16 / ksplit;
    const unsigned grid = (unsigned)((N + rows_per_wg - 1) / rows_per_wg);
#define GV2(MR, KSV) hipLaunchKernelGGL((gemv_kernel<MR, KSV>), dim3(grid), dim3(NT), 0, s, x, ldx, W, ldw, y, ldy, (int)M, (int)N, (int)K, bias, residual, ldr, flags)
#define GV(MR) do { if (ksplit == 4) GV2(MR, 4); else if (ksplit == 2) GV2(MR, 2); else GV2(MR, 1); } while (0)
    if (M == 1) GV(1);
    else if (M == 2) GV(2);
    else GV(4);
#undef GV2
#undef GV
    return mm_launch_status();
}

extern "C" int mm355_rope_kv_append(mm355_bf16* qkv, int64_t ld, int64_t B, int64_t Hq, int64_t Hkv, int64_t d, const mm355_bf16* cos_t,
                                    const mm355_bf16* sin_t, const int32_t* positions, mm355_bf16* k_cache, mm355_bf16* v_cache, int64_t ld_kv,
                                    int64_t batch_stride_kv, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!qkv || !cos_t || !sin_t || !positions || !k_cache || !v_cache || B <= 0 || Hq <= 0 || Hkv <= 0 || d <= 0 || (d & 15) || (ld & 7) ||
        (ld_kv & 7) || (batch_stride_kv & 7) || B > 65535)
        return MM355_EINVAL;
    const int64_t work = (Hq + Hkv) * (d / 16) + Hkv * d / 8;
    hipLaunchKernelGGL(rope_kv_append_kernel, dim3((unsigned)((work + NT - 1) / NT), (unsigned)B), dim3(NT), 0, (hipStream_t)stream, qkv, ld,
                       (int)Hq, (int)Hkv, (int)d, cos_t, sin_t, positions, k_cache, v_cache, ld_kv, batch_stride_kv);
    return mm_launch_status();
}

// the arrival counters (one per sample and KV head, Hkv <= Hq) sit at the START of the workspace, at an offset that does not depend on
// max_kv_len: one workspace serves calls with different cache lengths (the records behind them are scratch, rewritten by every launch)
static int64_t decode_counter_floats(int64_t B, int64_t Hq) { return (B * Hq + 3) & ~(int64_t)3; }

extern "C" int64_t mm355_attn_decode_ws_floats(int64_t B, int64_t Hq, int64_t d, int64_t max_kv_len) {
    if (B <= 0 || Hq <= 0 || d <= 0 || max_kv_len <= 0) return 0;
    return decode_counter_floats(B, Hq) + B * Hq * ((max_kv_len + CH - 1) / CH) * (d + 2);
}

static int attn_decode_impl(const mm355_bf16* q, int64_t ld_q, const mm355_bf16* k_cache, const mm355_bf16* v_cache, int64_t ld_kv,
                            int64_t batch_stride_kv, const int32_t* kv_lens, int64_t max_kv_len, mm355_bf16* o, int64_t ld_o, int64_t B,
                            int64_t Hq, int64_t Hkv, int64_t d, float scale, float* workspace, int variant, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!q || !k_cache || !v_cache || !kv_lens || !o || !workspace || B <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || max_kv_len <= 0)
        return MM355_EINVAL;
    if (d <= 0 || d > 128 || (d & 7) || (ld_kv & 7) || (batch_stride_kv & 7) || !mm_aligned16(k_cache) || !mm_aligned16(v_cache)) return MM355_EINVAL;
    if (B > 65535 || Hkv > 65535) return MM355_EINVAL;
    if (variant < 0 || variant > 2) return MM355_EINVAL;      // 2 = variant 0 with the whole GQA group in one workgroup whatever the bound
    const int64_t G0_ = Hq / Hkv;
    // (beyond 16 sequences every CU already holds two 1024-thread workgroups: spreading the group only multiplies the K / V loads and their
    // unpacking -- measured at 32 / 64 sequences of ~520 rows: 31.1 / 55.6 us spread against 21.2 / 39.0 with the group in one workgroup;
    // 16 sequences: 17.2 against 18.8.  Same arithmetic per head either way.)
    if (variant == 0 && max_kv_len <= 4 * CH && (G0_ == 2 || G0_ == 4 || G0_ == 8) && B * Hq <= 512) {
        // a cache bound of <= 1024 rows is ONE key group: its lone workgroup per head group finishes without records or fences, so the
        // GQA group can be spread over more workgroups -- one query head each (two from 64 workgroups on): 32 workgroups instead of 8 for
        // one LLaMA-3-8B sample, 21 -> 10.5 us per launch (profiles/r5_attn_decode_heads_per_workgroup.log); the K / V rows are read by
        // every workgroup of the group (L2 hits).  Same arithmetic per head: bit-identical outputs.  Longer bounds keep the whole group in
        // one workgroup (fewer fences when the groups merge: the split forms are 2 - 4 x SLOWER there).
        const int G0 = (int)(Hq / Hkv);
        const int gw = (B * Hq <= 64 || G0 == 2) ? 1 : 2;
        int* counters_ = (int*)workspace;
        float* recs_ = workspace + decode_counter_floats(B, Hq);
        hipStream_t s_ = (hipStream_t)stream;
        if (gw == 1) return launch_wide<1>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, recs_, 1, (int)B, (int)Hq, (int)Hq, (int)d, scale, o, ld_o, counters_, s_, G0);
        return launch_wide<2>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, recs_, 1, (int)B, (int)Hq, (int)(Hq / 2), (int)d, scale, o, ld_o, counters_, s_, G0 / 2);
    }
    const int G = (int)(Hq / Hkv);
    const int nsplit = (int)((max_kv_len + CH - 1) / CH);
    const int ngroup = (nsplit + 3) / 4;                     // 1024-key groups: one workgroup each
    hipStream_t s = (hipStream_t)stream;
    int* counters = (int*)workspace;                         // zero on first use (caller), left zero by every launch
    workspace += decode_counter_floats(B, Hq);               // the per-chunk / per-group records
    int rc;
#define AD(Gv) (variant == 1 ? launch_split<Gv>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, workspace, nsplit, (int)B, (int)Hq, (int)Hkv, (int)d, scale, o, ld_o, counters, s) \
                             : launch_wide<Gv>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, workspace, ngroup, (int)B, (int)Hq, (int)Hkv, (int)d, scale, o, ld_o, counters, s))
    switch (G) {
        case 1: rc = AD(1); break;
        case 2: rc = AD(2); break;
        case 4: rc = AD(4); break;
        case 8: rc = AD(8); break;
        default: return MM355_EUNSUPPORTED;                  // GQA group sizes 1, 2, 4, 8
    }
#undef AD
    return rc;
}

extern "C" int mm355_attn_decode(const mm355_bf16* q, int64_t ld_q, const mm355_bf16* k_cache, const mm355_bf16* v_cache, int64_t ld_kv,
                                 int64_t batch_stride_kv, const int32_t* kv_lens, int64_t max_kv_len, mm355_bf16* o, int64_t ld_o, int64_t B,
                                 int64_t Hq, int64_t Hkv, int64_t d, float scale, float* workspace, void* stream) {
    return attn_decode_impl(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, max_kv_len, o, ld_o, B, Hq, Hkv, d, scale, workspace, 0, stream);
}
extern "C" int mm355_attn_decode_variant(const mm355_bf16* q, int64_t ld_q, const mm355_bf16* k_cache, const mm355_bf16* v_cache, int64_t ld_kv,
                                         int64_t batch_stride_kv, const int32_t* kv_lens, int64_t max_kv_len, mm355_bf16* o, int64_t ld_o, int64_t B,
                                         int64_t Hq, int64_t Hkv, int64_t d, float scale, float* workspace, int variant, void* stream) {
    return attn_decode_impl(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, max_kv_len, o, ld_o, B, Hq, Hkv, d, scale, workspace, variant, stream);
}

namespace {
}  // namespace

extern "C" int mm355_gemv_swiglu_bf16(const mm355_bf16* x, int64_t ldx, const mm355_bf16* Wgu, int64_t ldw, mm355_bf16* act, int64_t ld_act,
                                      int64_t M, int64_t I, int64_t K, const mm355_bf16* norm_w, float eps, void* stream) {
    (void)hipGetLastError();
    if (!x || !Wgu || !act || M <= 0 || I <= 0 || K <= 0) return MM355_EINVAL;
    if (M > 32 || (I & 1)) return MM355_EUNSUPPORTED;
    if ((K & 7) || (ldx & 7) || (ldw & 7) || (ld_act & 1) || !mm_aligned16(x) || !mm_aligned16(Wgu) || (((uintptr_t)act) & 3u)) return MM355_EINVAL;
    if (norm_w && !mm_aligned16(norm_w)) return MM355_EINVAL;
    if (I > 0x3fffffff || K > 0x7fffffff) return MM355_EINVAL;
    GemvFusedArgs a = {};
    a.x = x; a.ldx = ldx; a.W = Wgu; a.ldw = ldw; a.M = (int)M; a.N = (int)(2 * I); a.K = (int)K; a.norm_w = norm_w; a.eps = eps;
    a.out = act; a.ld_out = ld_act; a.I = (int)I;
    if (M > 4) {
        if (!gemv_mfma_addressable(M, 2 * I, K, ldx, ldw)) return MM355_EUNSUPPORTED;
        GemvMfmaArgs g = {};
        g.f = a;
        return launch_gemv_mfma<1>(g, I / 2, norm_w != nullptr, (hipStream_t)stream);
    }
    if (gemv_deep_applies(M, 2 * I, K, ldx, ldw)) {
        GemvMfmaArgs g = {};
        g.f = a;
        return launch_gemv_deep<1>(g, I / 2, norm_w != nullptr, (hipStream_t)stream);
    }
    return MM355_EUNSUPPORTED;                               // a weight beyond 32-bit byte offsets: the unfused launch sequence
}

extern "C" int mm355_gemv_rope_append_bf16(const mm355_bf16* x, int64_t ldx, const mm355_bf16* Wqkv, int64_t ldw, mm355_bf16* qkv, int64_t ld_qkv,
                                           int64_t M, int64_t Hq, int64_t Hkv, int64_t d, int64_t K, const mm355_bf16* norm_w, float eps,
                                           const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* positions, mm355_bf16* k_cache,
                                           mm355_bf16* v_cache, int64_t ld_kv, int64_t batch_stride_kv, void* stream) {
    (void)hipGetLastError();
    if (!x || !Wqkv || !qkv || !cos_t || !sin_t || !positions || !k_cache || !v_cache || M <= 0 || Hq <= 0 || Hkv <= 0 || d <= 0 || K <= 0)
        return MM355_EINVAL;
    if (M > 16 || (d & 3)) return MM355_EUNSUPPORTED;        // rotation partners in pairs: d / 2 even
    if ((K & 7) || (ldx & 7) || (ldw & 7) || (ld_qkv & 1) || (ld_kv & 3) || (batch_stride_kv & 3) || !mm_aligned16(x) || !mm_aligned16(Wqkv) ||
        (((uintptr_t)qkv) & 3u) || (((uintptr_t)k_cache) & 7u) || (((uintptr_t)v_cache) & 7u))
        return MM355_EINVAL;
    if (norm_w && !mm_aligned16(norm_w)) return MM355_EINVAL;
    const int64_t N = (Hq + 2 * Hkv) * d;
    if (N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    GemvFusedArgs a = {};
    a.x = x; a.ldx = ldx; a.W = Wqkv; a.ldw = ldw; a.M = (int)M; a.N = (int)N; a.K = (int)K; a.norm_w = norm_w; a.eps = eps;
    a.out = qkv; a.ld_out = ld_qkv; a.Hq = (int)Hq; a.Hkv = (int)Hkv; a.d = (int)d;
    a.cos_t = cos_t; a.sin_t = sin_t; a.positions = positions; a.kc = k_cache; a.vc = v_cache; a.ld_kv = ld_kv; a.bs_kv = batch_stride_kv;
    if (M > 4) {
        if (!gemv_mfma_addressable(M, N, K, ldx, ldw)) return MM355_EUNSUPPORTED;
        GemvMfmaArgs g = {};
        g.f = a;
        return launch_gemv_mfma<2>(g, N / 4, norm_w != nullptr, (hipStream_t)stream);
    }
    if (gemv_deep_applies(M, N, K, ldx, ldw)) {
        GemvMfmaArgs g = {};
        g.f = a;
        return launch_gemv_deep<2>(g, N / 4, norm_w != nullptr, (hipStream_t)stream);
    }
    return MM355_EUNSUPPORTED;                               // a weight beyond 32-bit byte offsets: the unfused launch sequence
}
