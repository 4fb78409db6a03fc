// Probe: mm355_gemm_bf16 throughput on the step's shapes while a side-stream kernel holds N CUs, the way overlapped RCCL kernels will
// (a RCCL channel = one workgroup that cannot share a CU with a 512-thread / 246-VGPR GEMM workgroup and keeps that CU for the whole
// collective).  Thief: N workgroups x 256 threads with 64 KiB of LDS each (so no GEMM workgroup fits beside it), spinning for a fixed
// wall-clock time.  Links libmm355.so through its C ABI.     build: hipcc --offload-arch=gfx950 -O3 -I include gemm_cu_thief.hip -L metamorph_amd/lib -lmm355
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "mm355.h"

__global__ __launch_bounds__(256) void thief(long long ticks, int* sink) {
    __shared__ int lds[16384];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) { acc += lds[(acc + threadIdx.x) & 16383]; __builtin_amdgcn_s_sleep(8); }
    if (acc == 123456789) sink[0] = acc;
}

int main() {
    const int64_t M = 32768;
    struct Sh { const char* name; int64_t m, n, k; } shapes[] = {
        {"o_proj fwd", M, 4096, 4096}, {"qkv fwd", M, 6144, 4096}, {"gate_up fwd", M, 28672, 4096}, {"down fwd", M, 4096, 14336},
        {"dW down", 4096, 14336, M}, {"dW gate_up", 28672, 4096, M}};
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    int* sink; hipMalloc(&sink, 4);
    size_t maxel = (size_t)M * 28672;
    uint16_t *a, *b, *c;
    hipMalloc(&a, maxel * 2); hipMalloc(&b, maxel * 2); hipMalloc(&c, maxel * 2);
    std::vector<uint16_t> h(1 << 24);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3f00 + (x >> 20) % 256) | ((x >> 19) & 1 ? 0x8000 : 0); }   // +-[0.5, 1)
    for (size_t off = 0; off < maxel; off += h.size()) {
        size_t n = std::min(h.size(), maxel - off);
        hipMemcpy(a + off, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(b + off, h.data(), n * 2, hipMemcpyHostToDevice);
    }
    printf("wall clock %d kHz; thief = N workgroups x 64 KiB LDS spinning on a side stream\n", wall_khz);
    printf("%-14s %8s", "shape", "tiles");
    const int thieves[] = {0, 8, 16, 32, 64};
    for (int n : thieves) printf("   N=%-3d TF/s (ms)", n);
    printf("\n");
    for (auto& sh : shapes) {
        printf("%-14s %8lld", sh.name, (long long)(((sh.m + 255) / 256) * ((sh.n + 255) / 256)));
        for (int n : thieves) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int w = 0; w < 2; ++w) mm355_gemm_bf16(a, sh.k, b, sh.k, c, sh.n, sh.m, sh.n, sh.k, nullptr, nullptr, 0, 0, 0, 0, s1);
            hipDeviceSynchronize();
            const int reps = 6;
            if (n) {
                hipLaunchKernelGGL(thief, dim3(n), dim3(256), 0, s2, (long long)wall_khz * 120, sink);   // 120 ms
                hipStreamSynchronize(0);
                hipEventRecord(e0, s2);                     // make sure the thieves are resident before the GEMMs start
                for (volatile int spin = 0; spin < 2000000; ++spin) {}
            }
            hipEventRecord(e0, s1);
            for (int r = 0; r < reps; ++r) mm355_gemm_bf16(a, sh.k, b, sh.k, c, sh.n, sh.m, sh.n, sh.k, nullptr, nullptr, 0, 0, 0, 0, s1);
            hipEventRecord(e1, s1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            hipDeviceSynchronize();
            printf("   %8.1f (%5.2f)", 2.0 * sh.m * sh.n * sh.k / ms / 1e9, ms);
        }
        printf("\n");
    }
    return 0;
}
