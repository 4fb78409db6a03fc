// Issue rate of v_dot2c_f32_bf16 against v_fmac_f32 (one wave per SIMD, 16 independent accumulators): 5.24 vs 5.18 cycles per wave instruction
// on MI355X (v_dot2_f32_bf16 VOP3P: 5.65; v_dot2c_f32_f16: 5.17).  Build: hipcc --offload-arch=gfx950 -O3 dot2_rate_probe.hip -o dot2_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* o, uint32_t a, uint32_t b, long long* cyc) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.5f + i;
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 1024; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
            else if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
            else if (KIND == 2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            else asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    o[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
    float* o; long long* c; hipMalloc(&o, 1 << 20); hipMallocManaged(&c, 64);
    for (int wg = 1; wg <= 2; ++wg) {
        hipLaunchKernelGGL(k<0>, dim3(256 * wg), dim3(256), 0, 0, o, 1u, 2u, c);
        hipLaunchKernelGGL(k<1>, dim3(256 * wg), dim3(256), 0, 0, o, 1u, 2u, c);
        hipLaunchKernelGGL(k<2>, dim3(256 * wg), dim3(256), 0, 0, o, 1u, 2u, c);
        hipLaunchKernelGGL(k<3>, dim3(256 * wg), dim3(256), 0, 0, o, 1u, 2u, c);
        hipDeviceSynchronize();
        const char* nm[4] = {"v_dot2c_f32_bf16", "v_fmac_f32", "v_dot2_f32_bf16 (vop3p)", "v_dot2c_f32_f16"};
        for (int i = 0; i < 4; ++i) printf("%d WG/CU-ish: %-26s %.2f ticks(100MHz?) per 16384 ops per wave -> %.3f per op\n", wg, nm[i], (double)c[i], (double)c[i] / 16384);
    }
}
