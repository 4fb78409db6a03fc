// Decode-shape kernels (gfx950): the same decoder as the training path, but for a handful of new rows per step against a
// KV cache (greedy_decode / generate, reference metamorph_llama.py:502-597, 665-717, which re-runs the whole
// prefix every step with use_cache=False).  Everything here is HBM-bound streaming, not MFMA work:
//   * gemv_kernel      y[M,N] = x[M,K] W[N,K]^T for M <= 8: every weight row is read exactly once, 16 B per lane, four rows per
//                      wave in flight, fp32 accumulation, wave reduction, the usual bias / GELU / residual epilogue
//   * attn_decode_*    one query row per (sample, head) against [kv_len] cached keys: KV is split into 256-key chunks over
//                      workgroups (all query heads of a GQA group share one read of their K / V chunk), partial (max, sum, o)
//                      per chunk, merged by a second tiny kernel (flash-decoding)
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
                                                               int Hq, int Hkv, int d, float scale) {
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
    if (k0 >= kv_len) {                                      // chunk beyond the cache: empty partial
        if (tid < G) { float* w = wrec + (int64_t)tid * nsplit * rec; w[0] = -INFINITY; w[1] = 0.f; }
        return;
    }
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

__global__ __launch_bounds__(128) void attn_decode_merge_kernel(const float* __restrict__ ws, uint16_t* __restrict__ o, int64_t ld_o, int nsplit,
                                                                int Hq, int d) {
    const int hq = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
    const int rec = d + 2;
    const float* w = ws + ((int64_t)b * Hq + hq) * nsplit * rec;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, w[(int64_t)s * rec]);
    float l = 0.f, acc = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float m = w[(int64_t)s * rec];
        if (m == -INFINITY) continue;
        const float f = __expf(m - M);
        l += f * w[(int64_t)s * rec + 1];
        if (c < d) acc += f * w[(int64_t)s * rec + 2 + c];
    }
    if (c < d) o[(int64_t)b * ld_o + (int64_t)hq * d + c] = f2bf(l > 0.f ? acc / l : 0.f);
}

template <int G>
int launch_split(const uint16_t* q, int64_t ld_q, const uint16_t* kc, const uint16_t* vc, int64_t ld_kv, int64_t bs_kv, const int32_t* kv_lens,
                 float* ws, int nsplit, int B, int Hq, int Hkv, int d, float scale, hipStream_t s) {
    hipLaunchKernelGGL(attn_decode_split_kernel<G>, dim3((unsigned)nsplit, (unsigned)Hkv, (unsigned)B), dim3(NT), 0, s, q, ld_q, kc, vc, ld_kv,
                       bs_kv, kv_lens, ws, nsplit, Hq, Hkv, d, scale);
    return mm_launch_status();
}

}  // namespace

extern "C" int mm355_gemv_bf16(const mm355_bf16* x, int64_t ldx, const mm355_bf16* W, int64_t ldw, void* y, int64_t ldy, int64_t M, int64_t N,
                               int64_t K, const mm355_bf16* bias, const mm355_bf16* residual, int64_t ldr, uint32_t flags, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0) return MM355_EINVAL;
    if (M > 8) return MM355_EUNSUPPORTED;                    // more rows: mm355_gemm_bf16
    if ((K & 7) || (ldx & 7) || (ldw & 7) || !mm_aligned16(x) || !mm_aligned16(W)) return MM355_EINVAL;
    if ((flags & MM355_GEMM_BIAS) && !bias) return MM355_EINVAL;
    if ((flags & MM355_GEMM_RESIDUAL) && !residual) return MM355_EINVAL;
    if ((flags & MM355_GEMM_GELU_ERF) && (flags & MM355_GEMM_GELU_TANH)) return MM355_EINVAL;
    if (flags & MM355_GEMM_ACCUMULATE) return MM355_EUNSUPPORTED;
    if (N > 0x7fffffff || K > 0x7fffffff) return MM355_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // few rows of a long K (down projection): split K over the waves of a workgroup; measured: K = 14336, N = 4096 28.5 -> 24.3 us,
    // while at K = 4096 the extra LDS hand-over costs more than it buys (13.5 -> 17.7 us)
    const int ksplit = (N <= 8192 && K >= 8192) ? 4 : 1;
    const int rows_per_wg = 16 / ksplit;
    const unsigned grid = (unsigned)((N + rows_per_wg - 1) / rows_per_wg);
#define GV2(MR, KSV) hipLaunchKernelGGL((gemv_kernel<MR, KSV>), dim3(grid), dim3(NT), 0, s, x, ldx, W, ldw, y, ldy, (int)M, (int)N, (int)K, bias, residual, ldr, flags)
#define GV(MR) do { if (ksplit == 4) GV2(MR, 4); else if (ksplit == 2) GV2(MR, 2); else GV2(MR, 1); } while (0)
    if (M == 1) GV(1);
    else if (M == 2) GV(2);
    else if (M <= 4) GV(4);
    else GV(8);
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

extern "C" int64_t mm355_attn_decode_ws_floats(int64_t B, int64_t Hq, int64_t d, int64_t max_kv_len) {
    if (B <= 0 || Hq <= 0 || d <= 0 || max_kv_len <= 0) return 0;
    return B * Hq * ((max_kv_len + CH - 1) / CH) * (d + 2);
}

extern "C" int mm355_attn_decode(const mm355_bf16* q, int64_t ld_q, const mm355_bf16* k_cache, const mm355_bf16* v_cache, int64_t ld_kv,
                                 int64_t batch_stride_kv, const int32_t* kv_lens, int64_t max_kv_len, mm355_bf16* o, int64_t ld_o, int64_t B,
                                 int64_t Hq, int64_t Hkv, int64_t d, float scale, float* workspace, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!q || !k_cache || !v_cache || !kv_lens || !o || !workspace || B <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || max_kv_len <= 0)
        return MM355_EINVAL;
    if (d <= 0 || d > 128 || (d & 7) || (ld_kv & 7) || (batch_stride_kv & 7) || !mm_aligned16(k_cache) || !mm_aligned16(v_cache)) return MM355_EINVAL;
    if (B > 65535 || Hkv > 65535) return MM355_EINVAL;
    const int G = (int)(Hq / Hkv);
    const int nsplit = (int)((max_kv_len + CH - 1) / CH);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (G) {
        case 1: rc = launch_split<1>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, workspace, nsplit, (int)B, (int)Hq, (int)Hkv, (int)d, scale, s); break;
        case 2: rc = launch_split<2>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, workspace, nsplit, (int)B, (int)Hq, (int)Hkv, (int)d, scale, s); break;
        case 4: rc = launch_split<4>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, workspace, nsplit, (int)B, (int)Hq, (int)Hkv, (int)d, scale, s); break;
        case 8: rc = launch_split<8>(q, ld_q, k_cache, v_cache, ld_kv, batch_stride_kv, kv_lens, workspace, nsplit, (int)B, (int)Hq, (int)Hkv, (int)d, scale, s); break;
        default: return MM355_EUNSUPPORTED;                  // GQA group sizes 1, 2, 4, 8
    }
    if (rc != MM355_OK) return rc;
    hipLaunchKernelGGL(attn_decode_merge_kernel, dim3((unsigned)Hq, (unsigned)B), dim3(128), 0, s, workspace, o, ld_o, nsplit, (int)Hq, (int)d);
    return mm_launch_status();
}
