// Probe: what does ds_read_b64_tr_b16 return?  LDS is filled with lds[e] = e (16-bit element index);
// lane l reads from byte address addr_bytes[l]; prints the 4 returned 16-bit values per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__global__ void probe(const int* addr_bytes, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(uintptr_t)lds + (uint32_t)addr_bytes[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int *d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int test = 0; test < 3; ++test) {
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, i = l & 15, j = i >> 2, q = i & 3;
            if (test == 0) h_addr[l] = l * 8;                                   // linear: lane l -> elements 4l..4l+3
            else if (test == 1) h_addr[l] = ((g * 4 + j) * 128 + q * 4) * 2;      // rows of stride 128 el: row = g*4+j, col = q*4
            else h_addr[l] = ((g * 8 + j) * 256 + 32 + q * 4) * 2;               // stride 256 el, col base 32, rows g*8+j
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("test %d\n", test);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr_el %5d -> %5u %5u %5u %5u\n", l, h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
        }
    }
    return 0;
}
