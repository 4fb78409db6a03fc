// Probe: LDS cycles per wave-instruction for the fragment-read patterns of the attention kernels (ds_read_b64_tr_b16 gathers of V,
// ds_read_b128 rows of K) under different tile layouts / swizzles.  One workgroup of 4 waves (one per SIMD), each wave issues
// REPS x 16 reads; prints shader cycles per read instruction per wave (all 4 waves concurrently => CU-level LDS throughput).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int REPS = 256;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const int* off, long long* cycles, int* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int o[16];
    for (int i = 0; i < 16; ++i) o[i] = off[i * 64 + lane];
    int acc = 0;
    __syncthreads();
    long long t0 = clock64();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(lds + o[i]));
                acc += v[0] + v[3];
            } else {
                const u32x4 v = *(const u32x4*)(lds + o[i]);
                acc += v.x + v.w;
            }
        }
    }
    long long t1 = clock64();
    if (lane == 0) cycles[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = acc;
}

int main() {
    int h_off[16 * 64];
    int* d_off; long long* d_cyc; int* d_sink;
    hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_cyc, 4 * sizeof(long long)); hipMalloc(&d_sink, 256 * sizeof(int));
    const char* names[] = {
        "tr: attn3 V (row = fq*4+fr/4, 256-B rows, swz 2*(row&7))", "tr: 256-B rows, no swizzle", "tr: 256-B rows, quarter swizzle (row&3)<<2",
        "tr: 32x32 A-frag (d = l%32, hi = l/32), 256-B rows, swz 2*(row&7)", "tr: 32x32 A-frag, quarter swizzle (row&3)<<2", "tr: 32x32 A-frag, no swizzle",
        "tr: 128 contiguous B per 16-lane group, groups +512 B (guide layout 1)", "tr: fully linear 512 B",
        "b128: 16x16 K rows (row = l&15, chunk = 4kk + l/16), swz 2*(row&7)", "b128: 32x32 K rows (row = l%32, chunk = 2ks + l/32), swz 2*(row&7)",
        "b128: 32x32 K rows, swz row&15", "b128: 32x32 K rows, no swizzle", "b128: linear (lane*16)"};
    for (int pat = 0; pat < 13; ++pat) {
        for (int i = 0; i < 16; ++i)
            for (int l = 0; l < 64; ++l) {
                int fr = l & 15, fq = l >> 4, off = 0;
                if (pat <= 2) {                               // attn3: read i: kk = i / 8, j = i % 8 (d block of 16)
                    int kk = i >> 3, j = i & 7;
                    int row = kk * 32 + fq * 4 + (fr >> 2), chunk = j * 2 + ((fr & 3) >> 1), sub = (fr & 1) * 8;
                    int swz = pat == 0 ? 2 * (row & 7) : pat == 1 ? 0 : ((row & 3) << 2);
                    off = row * 256 + ((chunk ^ swz) << 4) + sub;
                } else if (pat <= 5) {                        // 32x32x16 A = V^T[d][key]: read i: db = i / 4 (32 d), j = i % 4 (16 keys)
                    int db = i >> 2, j = i & 3, g = l >> 4, mh = g & 1, hi = g >> 1, ii = l & 15;
                    int row = j * 16 + 4 * hi + (ii >> 2), col = db * 32 + mh * 16 + (ii & 3) * 4;
                    int chunk = col >> 3, sub = (col & 7) * 2;
                    int swz = pat == 3 ? 2 * (row & 7) : pat == 4 ? ((row & 3) << 2) : 0;
                    off = row * 256 + ((chunk ^ swz) << 4) + sub;
                } else if (pat == 6) {
                    off = ((l & 15) + (i & 3) * 16 + (l >> 4) * 64) * 8 + (i >> 2) * 4096;
                } else if (pat == 7) {
                    off = l * 8 + i * 512;
                } else if (pat == 8) {                        // 16x16x32 K operand: rows l&15 (+16 j), chunk kk*4 + fq
                    int kk = i >> 2, j = i & 3, row = j * 16 + fr, chunk = kk * 4 + fq;
                    off = row * 256 + ((chunk ^ (2 * (row & 7))) << 4);
                } else if (pat <= 11) {                       // 32x32x16 K operand: rows l%32 (+32 kb), chunk ks*2 + l/32
                    int kb = i >> 3, ks = i & 7, row = kb * 32 + (l & 31), chunk = ks * 2 + (l >> 5);
                    int swz = pat == 9 ? 2 * (row & 7) : pat == 10 ? (row & 15) : 0;
                    off = row * 256 + ((chunk ^ swz) << 4);
                } else {
                    off = l * 16 + i * 1024;
                }
                h_off[i * 64 + l] = off;
            }
        hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
        long long h_cyc[4];
        for (int rep = 0; rep < 2; ++rep) {
            if (pat <= 7) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(256), 0, 0, d_off, d_cyc, d_sink);
            else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(256), 0, 0, d_off, d_cyc, d_sink);
            hipMemcpy(h_cyc, d_cyc, sizeof(h_cyc), hipMemcpyDeviceToHost);
        }
        printf("%-75s %6.2f cyc / wave-instruction (4 waves concurrently)\n", names[pat], (double)h_cyc[0] / (REPS * 16));
    }
    return 0;
}
