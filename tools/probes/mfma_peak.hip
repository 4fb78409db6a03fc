// Practical MFMA ceiling probe: register-only v_mfma_f32_16x16x32_bf16 / 32x32x16 loops on every CU.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/probes/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> double timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e-3;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 512 * 4);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        for (int blocks : {256, 512}) {
            double t = timeit([&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
            double fl = (double)blocks * (threads / 64) * iters * 16 * 16384.0;
            printf("16x16x32 blocks=%d threads=%d: %.1f TF/s (%.2f ms)\n", blocks, threads, fl / t / 1e12, t * 1e3);
            t = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
            fl = (double)blocks * (threads / 64) * iters * 4 * 32768.0;
            printf("32x32x16 blocks=%d threads=%d: %.1f TF/s (%.2f ms)\n", blocks, threads, fl / t / 1e12, t * 1e3);
        }
    }
    // sustained: ~2 s of back-to-back launches, report the last
    double t = 0;
    for (int r = 0; r < 40; ++r) t = timeit([&] { hipLaunchKernelGGL(k16<16>, dim3(256), dim3(512), 0, 0, out, iters * 2); });
    printf("sustained 16x16x32: %.1f TF/s\n", 256.0 * 8 * iters * 2 * 16 * 16384.0 / t / 1e12);
    return 0;
}
