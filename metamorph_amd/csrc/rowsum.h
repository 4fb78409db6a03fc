// Deterministic reduction of per-row loss values (rowwise.hip, losses.hip, linear_ce.hip).
#pragma once
#include "mm355_common.h"

// Fixed-order sum of n floats by ONE workgroup of 1024 threads: thread t adds v[t], v[t + 1024], ... in that order, then a fixed tree over
// the 16 waves.  out[0] (+)= scale * sum.  The loss scalars (CE / cosine / soft-CE / mean-abs row values) go through this instead of one
// fp32 atomicAdd per row, so two runs on the same inputs print the same bits.
static __global__ __launch_bounds__(1024) void mm_sum_rows_kernel(const float* __restrict__ v, int64_t n, float scale, float* __restrict__ out,
                                                                  int accumulate) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += v[i];
    const float t = block_sum<1024>(acc, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * t;
}

