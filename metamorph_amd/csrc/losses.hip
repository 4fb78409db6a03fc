// Image-AR head variants next to the cosine loss of rowwise.hip (reference
// metamorph/model/language_model/metamorph_llama.py:433-459 and :211-219), one wave per row, HBM-bound streaming kernels:
//   mean-abs ("mse_loss_fn", the constructor default normalize_vision=False / apply_softmax=False),
//   soft cross-entropy against temperature-0.07 softmax targets (apply_softmax=True),
//   the temperature softmax itself (tower side siglip_encoder.py:210-211, decode side metamorph_llama.py:372-373) + backward.
// Rows are short (C = mm_hidden_size = 1152): every pass re-reads the row from L1/L2, 16 B per lane.
#include "mm355_common.h"
#include "rowsum.h"

namespace {

constexpr int NT = 256;           // 4 rows per workgroup

MM_DEV float signf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// ------------------------------------------------------------------------------------------------
// mean |target - pred| over all R*C elements (the reference's per-row mean of |z_i - h_i|, averaged over rows).
// abs_sum += sum |round_bf(t - p)|;  dpred = sign(p - t) / (R*C)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void mean_abs_loss_kernel(const uint16_t* __restrict__ pred, const uint16_t* __restrict__ tgt, int R, int C,
                                                           float* __restrict__ abs_sum, uint16_t* __restrict__ dpred, float* __restrict__ row_out) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int nv = C >> 3;
    const uint16_t* pr = pred + (int64_t)row * C;
    const uint16_t* tr = tgt + (int64_t)row * C;
    uint16_t* gr = dpred ? dpred + (int64_t)row * C : nullptr;
    const float gs = 1.0f / ((float)R * (float)C);
    float acc = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float p[8], t[8], g[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
        unpack8(*(const u32x4*)(tr + v * 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = round_bf(t[e] - p[e]);           // the bf16 subtraction of the reference stack
            acc += fabsf(d);
            g[e] = -signf(d) * gs;
        }
        if (gr) *(u32x4*)(gr + v * 8) = pack8(g);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (row_out) row_out[row] = acc;                    // summed in a fixed order afterwards (mm_sum_rows_kernel)
        else atomicAdd(abs_sum, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// Temperature softmax of a row:  y = softmax(round_bf(x * inv_temp))  (bf16 in, bf16 out; fp32 inside like torch).
// ------------------------------------------------------------------------------------------------
MM_DEV void row_softmax_stats(const uint16_t* xr, int nv, int lane, float inv_temp, float& mx, float& sum) {
    float m = -INFINITY;
    for (int v = lane; v < nv; v += 64) {
        float x[8];
        unpack8(*(const u32x4*)(xr + v * 8), x);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, round_bf(x[e] * inv_temp));
    }
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float x[8];
        unpack8(*(const u32x4*)(xr + v * 8), x);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf(round_bf(x[e] * inv_temp) - m);
    }
    mx = m;
    sum = wave_sum(s);
}

__global__ __launch_bounds__(NT) void softmax_rows_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int R, int C, float inv_temp) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int nv = C >> 3;
    const uint16_t* xr = x + (int64_t)row * C;
    uint16_t* yr = y + (int64_t)row * C;
    float m, s;
    row_softmax_stats(xr, nv, lane, inv_temp, m, s);
    const float inv = 1.0f / s;
    for (int v = lane; v < nv; v += 64) {
        float f[8];
        unpack8(*(const u32x4*)(xr + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __expf(round_bf(f[e] * inv_temp) - m) * inv;
        *(u32x4*)(yr + v * 8) = pack8(f);
    }
}

// dx = y * (dy - sum_j dy_j y_j) * inv_temp   (y = the saved softmax output)
__global__ __launch_bounds__(NT) void softmax_rows_bwd_kernel(const uint16_t* __restrict__ y, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
                                                              int R, int C, float inv_temp) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int nv = C >> 3;
    const uint16_t* yr = y + (int64_t)row * C;
    const uint16_t* gr = dy + (int64_t)row * C;
    uint16_t* dr = dx + (int64_t)row * C;
    float dot = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float a[8], g[8];
        unpack8(*(const u32x4*)(yr + v * 8), a);
        unpack8(*(const u32x4*)(gr + v * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dot += a[e] * g[e];
    }
    dot = wave_sum(dot);
    for (int v = lane; v < nv; v += 64) {
        float a[8], g[8];
        unpack8(*(const u32x4*)(yr + v * 8), a);
        unpack8(*(const u32x4*)(gr + v * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = a[e] * (g[e] - dot) * inv_temp;
        *(u32x4*)(dr + v * 8) = pack8(g);
    }
}

// ------------------------------------------------------------------------------------------------
// Soft cross-entropy head (apply_softmax=True):
//   u = normalize ? F.normalize(pred_raw) (bf16) : pred_raw;   q = softmax(round_bf(u / 0.07)) (bf16)
//   loss_sum += -sum_j target_j * log(q_j + 1e-10);   dpred = d(mean_r loss_r) / d pred_raw
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void soft_ce_loss_kernel(const uint16_t* __restrict__ pred, const uint16_t* __restrict__ tgt, int R, int C,
                                                          int normalize, float inv_temp, float* __restrict__ loss_sum, uint16_t* __restrict__ dpred,
                                                          float* __restrict__ row_out) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int nv = C >> 3;
    const uint16_t* pr = pred + (int64_t)row * C;
    const uint16_t* tr = tgt + (int64_t)row * C;
    float pp = 0.f;
    if (normalize) {
        for (int v = lane; v < nv; v += 64) {
            float p[8];
            unpack8(*(const u32x4*)(pr + v * 8), p);
#pragma unroll
            for (int e = 0; e < 8; ++e) pp += p[e] * p[e];
        }
        pp = wave_sum(pp);
    }
    const float pn = normalize ? fmaxf(round_bf(sqrtf(pp)), 1e-12f) : 1.0f;   // F.normalize on a bf16 tensor
    // z_j = round_bf(u_j * inv_temp), u_j = round_bf(p_j / pn)
    float m = -INFINITY;
    for (int v = lane; v < nv; v += 64) {
        float p[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = normalize ? round_bf(p[e] / pn) : p[e];
            m = fmaxf(m, round_bf(u * inv_temp));
        }
    }
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float p[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = normalize ? round_bf(p[e] / pn) : p[e];
            s += __expf(round_bf(u * inv_temp) - m);
        }
    }
    s = wave_sum(s);
    const float inv_s = 1.0f / s;
    // loss and  gq = sum_j g_j q_j  with g_j = d loss / d q_j = -t_j / (q_j + eps)
    float loss = 0.f, gq = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float p[8], t[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
        unpack8(*(const u32x4*)(tr + v * 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = normalize ? round_bf(p[e] / pn) : p[e];
            const float q = round_bf(__expf(round_bf(u * inv_temp) - m) * inv_s);
            loss -= t[e] * __logf(q + 1e-10f);
            gq -= t[e] * q / (q + 1e-10f);
        }
    }
    loss = wave_sum(loss);
    gq = wave_sum(gq);
    if (lane == 0) {
        if (row_out) row_out[row] = loss;
        else atomicAdd(loss_sum, loss);
    }
    if (!dpred) return;
    uint16_t* gr = dpred + (int64_t)row * C;
    const float rs = inv_temp / (float)R;                      // mean over rows, d z / d u
    // du_k = q_k (g_k - gq) * rs ;  normalised: dp = (du - uhat (uhat . du)) / |p|
    float ud = 0.f;
    const float pnorm = fmaxf(sqrtf(pp), 1e-20f);
    if (normalize) {
        for (int v = lane; v < nv; v += 64) {
            float p[8], t[8];
            unpack8(*(const u32x4*)(pr + v * 8), p);
            unpack8(*(const u32x4*)(tr + v * 8), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = round_bf(p[e] / pn);
                const float q = round_bf(__expf(round_bf(u * inv_temp) - m) * inv_s);
                const float du = q * (-t[e] / (q + 1e-10f) - gq) * rs;
                ud += (p[e] / pnorm) * du;
            }
        }
        ud = wave_sum(ud);
    }
    for (int v = lane; v < nv; v += 64) {
        float p[8], t[8], g[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
        unpack8(*(const u32x4*)(tr + v * 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = normalize ? round_bf(p[e] / pn) : p[e];
            const float q = round_bf(__expf(round_bf(u * inv_temp) - m) * inv_s);
            const float du = q * (-t[e] / (q + 1e-10f) - gq) * rs;
            g[e] = normalize ? (du - (p[e] / pnorm) * ud) / pnorm : du;
        }
        *(u32x4*)(gr + v * 8) = pack8(g);
    }
}

inline unsigned row_grid(int64_t R) { return (unsigned)((R + NT / 64 - 1) / (NT / 64)); }

}  // namespace

extern "C" int mm355_mean_abs_loss(const mm355_bf16* pred, const mm355_bf16* target, int64_t R, int64_t C, float* abs_sum,
                                   mm355_bf16* dpred, float* row_ws, void* stream) {
    (void)hipGetLastError();
    if (!pred || !target || !abs_sum || R <= 0 || C <= 0 || (C & 7) || R > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(mean_abs_loss_kernel, dim3(row_grid(R)), dim3(NT), 0, (hipStream_t)stream, pred, target, (int)R, (int)C, abs_sum, dpred, row_ws);
    if (row_ws) hipLaunchKernelGGL(mm_sum_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)row_ws, R, 1.0f, abs_sum, 1);
    return mm_launch_status();
}

extern "C" int mm355_soft_ce_loss(const mm355_bf16* pred_raw, const mm355_bf16* target, int64_t R, int64_t C, int normalize, float temperature,
                                  float* loss_sum, mm355_bf16* dpred, float* row_ws, void* stream) {
    (void)hipGetLastError();
    if (!pred_raw || !target || !loss_sum || R <= 0 || C <= 0 || (C & 7) || R > 0x7fffffff || !(temperature > 0.f)) return MM355_EINVAL;
    hipLaunchKernelGGL(soft_ce_loss_kernel, dim3(row_grid(R)), dim3(NT), 0, (hipStream_t)stream, pred_raw, target, (int)R, (int)C, normalize,
                       1.0f / temperature, loss_sum, dpred, row_ws);
    if (row_ws) hipLaunchKernelGGL(mm_sum_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)row_ws, R, 1.0f, loss_sum, 1);
    return mm_launch_status();
}

extern "C" int mm355_softmax_rows(const mm355_bf16* x, mm355_bf16* y, int64_t R, int64_t C, float temperature, void* stream) {
    (void)hipGetLastError();
    if (!x || !y || R <= 0 || C <= 0 || (C & 7) || R > 0x7fffffff || !(temperature > 0.f)) return MM355_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(row_grid(R)), dim3(NT), 0, (hipStream_t)stream, x, y, (int)R, (int)C, 1.0f / temperature);
    return mm_launch_status();
}

extern "C" int mm355_softmax_rows_bwd(const mm355_bf16* y, const mm355_bf16* dy, mm355_bf16* dx, int64_t R, int64_t C, float temperature,
                                      void* stream) {
    (void)hipGetLastError();
    if (!y || !dy || !dx || R <= 0 || C <= 0 || (C & 7) || R > 0x7fffffff || !(temperature > 0.f)) return MM355_EINVAL;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(row_grid(R)), dim3(NT), 0, (hipStream_t)stream, y, dy, dx, (int)R, (int)C, 1.0f / temperature);
    return mm_launch_status();
}
