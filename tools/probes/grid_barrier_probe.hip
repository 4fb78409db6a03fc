// Cost of a grid-wide barrier among co-resident workgroups on MI355X (one workgroup per CU, 256 workgroups), and what a persistent
// "weight stream with barriers" sustains: every phase streams BYTES per workgroup from a large buffer (16-B loads, 8 in flight per thread),
// then all workgroups meet.  Barrier kinds: 0 = none (upper bound of the stream), 1 = one counter per phase (agent-scope atomics),
// 2 = per-XCD counters + one global (workgroup ids go round-robin over the eight XCDs), 3 = kind 1 with the NEXT phase's first loads issued
// before the barrier (the stream continues while the workgroups wait).  Every spin is bounded (a bug cannot hang the GPU).
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier_probe.hip -o grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define SPIN_CAP (1 << 22)

__device__ __forceinline__ bool barrier_flat(int* cnt, int nwg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
            if (++spins > SPIN_CAP) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}
__device__ __forceinline__ bool barrier_hier(int* cx, int* cg, int nwg) {      // cx[8], cg[1]
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        const int xcd = blockIdx.x & 7, per = nwg >> 3;
        const int old = __hip_atomic_fetch_add(cx + xcd * 16, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == per - 1) __hip_atomic_fetch_add(cg, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cg, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 8) {
            if (++spins > SPIN_CAP) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

template <int KIND>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ w, long long nvec, int phases, const int* __restrict__ vec_per_wg,
                                                     int* cnt, unsigned* out, int* err) {
    const int nwg = gridDim.x;
    unsigned acc = 0;
    long long base = 0;
    u32x4 pre[8];
    bool have_pre = false;
    for (int p = 0; p < phases; ++p) {
        const int n = vec_per_wg[p % 5];                     // 16-B vectors per workgroup in this phase (multiple of 256 * 8)
        const long long off = (base + (long long)blockIdx.x * n) % (nvec - n);
        const u32x4* src = w + off + threadIdx.x;
        int i = 0;
        if (KIND == 3 && have_pre) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += pre[u].x ^ pre[u].w;
            i = 256 * 8;
        }
        for (; i < n; i += 256 * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].w;
        }
        base += (long long)nwg * n;
        if (KIND == 3 && p + 1 < phases) {                   // the next phase's first block, in flight across the barrier
            const int n2 = vec_per_wg[(p + 1) % 5];
            const long long off2 = (base + (long long)blockIdx.x * n2) % (nvec - n2);
#pragma unroll
            for (int u = 0; u < 8; ++u) pre[u] = __builtin_nontemporal_load(w + off2 + threadIdx.x + u * 256);
            have_pre = true;
        }
        bool ok = true;
        if (KIND == 1 || KIND == 3) ok = barrier_flat(cnt + p * 16, nwg);
        else if (KIND == 2) ok = barrier_hier(cnt + (long long)p * 16 * 9, cnt + (long long)p * 16 * 9 + 8 * 16, nwg);
        if (!ok) { if (threadIdx.x == 0) atomicAdd(err, 1); break; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const int phases = 160;
    // LLaMA-3-8B layer at one sequence: q|k|v 50 MB, attention (tiny), o 33.5 MB, gate|up 235 MB, down 117 MB
    const double mb[5] = {50.3, 0.5, 33.5, 234.9, 117.4};
    std::vector<int> vpw(5);
    double total = 0;
    for (int i = 0; i < 5; ++i) { long long v = (long long)(mb[i] * 1e6 / 16 / nwg); v = (v + 2047) / 2048 * 2048; vpw[i] = (int)v; total += (double)v * 16 * nwg; }
    const long long nvec = (4ll << 30) / 16;
    u32x4* w; hipMalloc(&w, nvec * 16); hipMemset(w, 1, nvec * 16);
    int *d_vpw, *cnt, *err; unsigned* out;
    hipMalloc(&d_vpw, 20); hipMemcpy(d_vpw, vpw.data(), 20, hipMemcpyHostToDevice);
    hipMalloc(&cnt, phases * 16 * 9 * 4); hipMalloc(&err, 4); hipMalloc(&out, nwg * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"no barrier", "one counter per phase", "per-XCD counters + global", "one counter + next phase's first loads across the barrier"};
    for (int kind = 0; kind < 4; ++kind) {
        float best = 1e9;
        int herr = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(cnt, 0, phases * 16 * 9 * 4); hipMemset(err, 0, 4);
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(nwg), dim3(256), 0, 0, w, nvec, phases, d_vpw, cnt, out, err);
            if (kind == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(nwg), dim3(256), 0, 0, w, nvec, phases, d_vpw, cnt, out, err);
            if (kind == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(nwg), dim3(256), 0, 0, w, nvec, phases, d_vpw, cnt, out, err);
            if (kind == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(nwg), dim3(256), 0, 0, w, nvec, phases, d_vpw, cnt, out, err);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        }
        const double bytes = total * (phases / 5);
        printf("%d workgroups, %-58s: %.3f ms for %d phases (%.1f GB) = %.2f TB/s, %.2f us per phase%s\n", nwg, names[kind], best, phases, bytes / 1e9,
               bytes / best / 1e9, best * 1e3 / phases, herr ? "  [SPIN CAP HIT]" : "");
    }
    return 0;
}
