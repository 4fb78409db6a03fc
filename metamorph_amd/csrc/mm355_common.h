// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libmm355.so.
// wave = 64 lanes; bf16 is carried as raw 16-bit words (uint16_t / short vectors).
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include "../../include/mm355.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;     // one MFMA A/B fragment (8 bf16 = 16 B)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;      // one 16x16 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MM_DEV __device__ __forceinline__

MM_DEV float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
MM_DEV uint16_t f2bf(float f) {                      // RNE, lowers to v_cvt_pk_bf16_f32 on gfx950
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
// one v_cvt_pk_bf16_f32 (the scalar form costs two conversions, a shift and an or)
typedef float mm_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mm_bf16x2 __attribute__((ext_vector_type(2)));
MM_DEV uint32_t pack2bf(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(mm_f32x2{lo, hi}, mm_bf16x2));
}
MM_DEV float bflo(uint32_t w) { return __uint_as_float(w << 16); }
MM_DEV float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
MM_DEV float round_bf(float f) { return bf2f(f2bf(f)); }

// 8 bf16 <-> 8 floats
MM_DEV void unpack8(const u32x4& v, float* f) {
    f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
    f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
MM_DEV u32x4 pack8(const float* f) {
    u32x4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

MM_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MM_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x = NT (multiple of 64); `red` needs NT/64 floats of LDS.
template <int NT>
MM_DEV float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
template <int NT>
MM_DEV float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}

MM_DEV float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
MM_DEV float gelu_tanh_f(float x) {
    const float c = 0.79788456080286535588f;
    return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}
MM_DEV float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
MM_DEV float gelu_tanh_grad(float x) {
    const float c = 0.79788456080286535588f;
    const float u = c * (x + 0.044715f * x * x * x);
    const float t = tanhf(u);
    const float du = c * (1.0f + 3.0f * 0.044715f * x * x);
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

static inline int mm_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MM355_OK : MM355_ELAUNCH;
}
// Dynamic-LDS opt-in of a kernel (> 64 KiB): hipFuncSetAttribute acts on the CURRENT device's copy of the function, so it is done once
// per device (a bit per device ordinal in a per-kernel mask); a process driving several GPUs gets every device configured.
static inline int mm_ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MM355_ELAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return MM355_ELAUNCH;
        done.fetch_or(bit, std::memory_order_release);       // idempotent: a concurrent first call just sets the attribute twice
    }
    return MM355_OK;
}
static inline bool mm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
