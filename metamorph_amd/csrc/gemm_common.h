// Shared between the GEMM translation units (gemm_bf16.hip, gemm_st.hip): the argument block and the fused store of eight adjacent
// output columns (bias / GELU / residual / accumulate / fp32 or bf16 output), so that every kernel rounds exactly the same way.
#pragma once
#include "mm355_common.h"

namespace {

struct GemmArgs {
    const uint16_t* A;
    const uint16_t* B;
    void* C;
    const uint16_t* bias;
    const uint16_t* res;
    int64_t lda, ldb, ldc, ldr, res_mod;
    int M, N, K;
    uint32_t flags;
    int ntm, ntn;
    int gm;                                                  // raster group height in tiles (ping-pong kernel)
    uint16_t* aux0;                                          // fused SwiGLU-backward epilogue: actT [I][ld_aux]
    uint16_t* aux1;                                          //                                 dguT [2 I][ld_aux]
    int64_t ld_aux;
    int kslice;                                              // split-K (gemm_nt_kernel only): K elements per blockIdx.y slice, 0 = off; C is then the
};                                                           // fp32 partial buffer [slices][M][ldc]

// ragged-edge epilogue (N tail or unaligned leading dimensions): one element at a time, kept out of line
// so the unrolled fast path stays small.
// (arguments by value: taking the address of the kernel-argument struct would push it into scratch memory)
__device__ __attribute__((noinline)) void epi_scalar(void* C, int64_t ldc, const uint16_t* bias, const uint16_t* res, int64_t ldr,
                                                     uint32_t fl, int N, int grow, int c, int64_t rr, float v0, float v1,
                                                     float v2, float v3, float v4, float v5, float v6, float v7) {
    const float v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
    for (int e = 0; e < 8; ++e) {
        const int ce = c + e;
        if (ce >= N) break;
        float x = v[e];
        if (fl & MM355_GEMM_BIAS) x += bf2f(bias[ce]);
        if (fl & MM355_GEMM_GELU_ERF) x = gelu_erf_f(x);
        else if (fl & MM355_GEMM_GELU_TANH) x = gelu_tanh_f(x);
        if (fl & MM355_GEMM_RESIDUAL) x += bf2f(res[rr * ldr + ce]);
        if (fl & MM355_GEMM_OUT_F32) {
            float* p = (float*)C + (int64_t)grow * ldc + ce;
            if (fl & MM355_GEMM_ACCUMULATE) x += *p;
            *p = x;
        } else {
            uint16_t* p = (uint16_t*)C + (int64_t)grow * ldc + ce;
            if (fl & MM355_GEMM_ACCUMULATE) x += bf2f(*p);
            *p = f2bf(x);
        }
    }
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// eight adjacent columns c .. c + 7 of output row `grow` (rr = residual row) from fp32 accumulators v[0..7]; c < N
MM_DEV void epi_store8(const GemmArgs& a, const uint32_t fl, const bool vec_ok, const int grow, const int64_t rr, const int c, float (&v)[8]) {
    const int N = a.N;
    uint16_t* Cb = (uint16_t*)a.C;
    float* Cf = (float*)a.C;
    const bool full = (c + 8 <= N) && vec_ok;
    if (full) {
        if (fl & MM355_GEMM_BIAS) {
            float b[8];
            unpack8(*(const u32x4*)(a.bias + c), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
        }
        if (fl & MM355_GEMM_GELU_ERF) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
        } else if (fl & MM355_GEMM_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
        }
        if (fl & MM355_GEMM_RESIDUAL) {
            float b[8];
            unpack8(*(const u32x4*)(a.res + rr * a.ldr + c), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
        }
        if (fl & MM355_GEMM_OUT_F32) {
            float* p = Cf + (int64_t)grow * a.ldc + c;
            if (fl & MM355_GEMM_ACCUMULATE) {
                const f32x4 o0 = *(const f32x4*)p, o1 = *(const f32x4*)(p + 4);
                v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w;
                v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w;
            }
            *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            uint16_t* p = Cb + (int64_t)grow * a.ldc + c;
            if (fl & MM355_GEMM_ACCUMULATE) {
                float b[8];
                unpack8(*(const u32x4*)p, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += b[e];
            }
            *(u32x4*)p = pack8(v);
        }
    } else {
        epi_scalar(a.C, a.ldc, a.bias, a.res, a.ldr, fl, N, grow, c, rr, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
}

}  // namespace
