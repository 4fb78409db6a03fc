// Probe: CU-level LDS throughput (LDS cycles per wave-instruction) of the fragment-read patterns of the attention kernels.
// One workgroup of NW waves on one CU; every wave issues REPS x 16 independent reads (inline asm: nothing is hoisted or merged),
// 16 in flight, then s_waitcnt.  cycles / (REPS * 16 * NW) = LDS cycles per wave-instruction when the LDS pipe is the limiter
// (NW = 8) -- compare NW = 1 (issue/latency bound) to see which regime a number is in.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
constexpr int REPS = 512;

template <int MODE>   // 0: ds_read_b128, 1: ds_read_b64_tr_b16, 2: ds_read_b64
__global__ __launch_bounds__(512) void probe(const int* off, long long* cycles, int* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t o[16];
    for (int i = 0; i < 16; ++i) o[i] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + off[i * 64 + lane];
    int acc = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {
                __attribute__((ext_vector_type(4))) uint32_t v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(o[i]));
                asm volatile("" :: "v"(v));
            } else if (MODE == 1) {
                __attribute__((ext_vector_type(2))) uint32_t v;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(o[i]));
                asm volatile("" :: "v"(v));
            } else {
                __attribute__((ext_vector_type(2))) uint32_t v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(o[i]));
                asm volatile("" :: "v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    __syncthreads();
    if (lane == 0) cycles[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = acc;
}

struct Pat { const char* name; int mode; int id; };

static int pattern(int id, int i, int l) {
    const int fr = l & 15, fq = l >> 4;
    switch (id) {
    case 0: {   // attn3 K rows, b128: row = fr (+16 j), chunk = kk*4 + fq, swz 2*(row&7)        (i: kk = i/4, j = i%4)
        const int kk = i >> 2, j = i & 3, row = j * 16 + fr, chunk = kk * 4 + fq;
        return row * 256 + ((chunk ^ (2 * (row & 7))) << 4);
    }
    case 1: {   // same rows, guide swizzle (row & 15)
        const int kk = i >> 2, j = i & 3, row = j * 16 + fr, chunk = kk * 4 + fq;
        return row * 256 + ((chunk ^ (row & 15)) << 4);
    }
    case 2: {   // same rows, no swizzle
        const int kk = i >> 2, j = i & 3, row = j * 16 + fr, chunk = kk * 4 + fq;
        return row * 256 + (chunk << 4);
    }
    case 3: return i * 1024 + l * 16;                     // b128 linear
    case 4: {   // attn3 V gathers, tr_b64: row = kk*32 + fq*4 + fr/4 (+16), chunk = j*2 + (fr&3)/2, sub = (fr&1)*8, swz 2*(row&7)
        const int kk = i >> 3, j = i & 7, row = kk * 32 + fq * 4 + (fr >> 2), chunk = j * 2 + ((fr & 3) >> 1);
        return row * 256 + ((chunk ^ (2 * (row & 7))) << 4) + (fr & 1) * 8;
    }
    case 5: {   // V gathers without swizzle
        const int kk = i >> 3, j = i & 7, row = kk * 32 + fq * 4 + (fr >> 2), chunk = j * 2 + ((fr & 3) >> 1);
        return row * 256 + (chunk << 4) + (fr & 1) * 8;
    }
    case 6: return i * 512 + l * 8;                       // 8 B per lane, linear
    case 7: {   // V gathers, rows of 16 lanes contiguous (128 B per 16-lane group, groups on consecutive rows)
        const int row = (i & 3) * 16 + fq * 4 + (i >> 2 & 1) * 2, c = fr * 8;
        return row * 256 + c + (i >> 3) * 128;
    }
    }
    return 0;
}

int main() {
    const Pat pats[] = {
        {"b128  attn3 K rows (row = fr, chunk = 4kk+fq), swz 2*(row&7)", 0, 0},
        {"b128  same rows, swz row&15", 0, 1},
        {"b128  same rows, no swizzle", 0, 2},
        {"b128  linear (lane*16)", 0, 3},
        {"tr64  attn3 V gathers, swz 2*(row&7)", 1, 4},
        {"tr64  attn3 V gathers, no swizzle", 1, 5},
        {"tr64  linear (lane*8)", 1, 6},
        {"tr64  128 contiguous B per 16 lanes", 1, 7},
        {"b64   attn3 V gather addresses (plain b64)", 2, 4},
        {"b64   linear (lane*8)", 2, 6},
    };
    int h_off[16 * 64];
    int* d_off; long long* d_cyc; int* d_sink;
    hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_cyc, 8 * sizeof(long long)); hipMalloc(&d_sink, 512 * sizeof(int));
    for (const Pat& p : pats) {
        for (int i = 0; i < 16; ++i)
            for (int l = 0; l < 64; ++l) h_off[i * 64 + l] = pattern(p.id, i, l);
        hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
        printf("%-62s", p.name);
        for (int nw : {1, 4, 8}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (p.mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64 * nw), 0, 0, d_off, d_cyc, d_sink);
                else if (p.mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64 * nw), 0, 0, d_off, d_cyc, d_sink);
                else hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64 * nw), 0, 0, d_off, d_cyc, d_sink);
                hipDeviceSynchronize();
            }
            long long c[8];
            hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
            long long mx = 0;
            for (int w = 0; w < nw; ++w) mx = c[w] > mx ? c[w] : mx;
            printf("  NW=%d: %6.2f cyc/instr/CU (%5.1f per wave)", nw, (double)mx / (REPS * 16.0 * nw), (double)mx / (REPS * 16.0));
        }
        printf("\n");
    }
    return 0;
}
