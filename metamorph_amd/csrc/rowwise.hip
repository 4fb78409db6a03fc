// Row-wise HBM-bound kernels of the MetaMorph hot path (gfx950): RMSNorm fwd/bwd, LayerNorm fwd,
// cross-entropy over logits rows, cosine regression loss, bilinear token reduction + L2 normalise.
// All loads/stores are 16-byte vectors (8 bf16); reductions are wave shuffles + one LDS hop.
#include "mm355_common.h"
#include "rowsum.h"
#include <type_traits>

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------------------
// RMSNorm forward: y = w * bf16(x * rsqrt(mean(x^2) + eps))      (HF LlamaRMSNorm order of roundings)
// one workgroup per row, row cached in registers (VPT vectors of 8 per thread).
// ------------------------------------------------------------------------------------------------
template <int VPT>
__global__ __launch_bounds__(NT) void rmsnorm_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                         uint16_t* __restrict__ y, float* __restrict__ rstd_out, int h, float eps) {
    __shared__ float red[NT / 64];
    const int64_t row = blockIdx.x;
    const int nv = h >> 3;
    const uint16_t* xr = x + row * h;
    float xv[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nv) {
            unpack8(*(const u32x4*)(xr + v * 8), xv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += xv[i][e] * xv[i][e];
        }
    }
    ss = block_sum<NT>(ss, red);
    const float rstd = rsqrtf(ss / (float)h + eps);
    if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nv) {
            float wv[8], o[8];
            unpack8(*(const u32x4*)(w + v * 8), wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = wv[e] * round_bf(xv[i][e] * rstd);
            *(u32x4*)(y + row * h + v * 8) = pack8(o);
        }
    }
}

// y^T of the same RMSNorm from the SAVED per-row rstd: out_t[c][r] = w[c] * bf16(x[r][c] * rstd[r]) -- the contraction-major operand the
// weight-gradient GEMMs of qkv / gate_up want.  One pass (read x, write y^T) instead of recomputing y and transposing it (two reads, two
// writes); bit-identical to transposing rmsnorm_fwd's output.  64 x 64 tiles through LDS, 16-B accesses on both sides.
__global__ __launch_bounds__(256) void rmsnorm_apply_t_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                              const float* __restrict__ rstd, int M, int h,
                                                              uint16_t* __restrict__ out, int64_t ld_out) {
    __shared__ uint16_t tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256, r = v >> 3, c = (v & 7) * 8;
        const int gr = r0 + r, gc = c0 + c;
        uint16_t tmp[8];
        if (gr < M && gc + 8 <= h) {
            float xv[8], wv[8], o[8];
            unpack8(*(const u32x4*)(x + (int64_t)gr * h + gc), xv);
            unpack8(*(const u32x4*)(w + gc), wv);
            const float rs = rstd[gr];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = wv[e] * round_bf(xv[e] * rs);
            *(u32x4*)tmp = pack8(o);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = 0;          // (h % 8 == 0: a vector is inside or outside as a whole)
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][c + e] = tmp[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256, c = v >> 3, r = (v & 7) * 8;   // output row = input col c
        const int gc = c0 + c, gr = r0 + r;
        if (gc >= h) continue;
        uint16_t tmp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tmp[e] = tile[r + e][c];
        if (gr + 8 <= M && ((ld_out & 7) == 0)) {
            *(u32x4*)(out + (int64_t)gc * ld_out + gr) = *(const u32x4*)tmp;
        } else {
            for (int e = 0; e < 8; ++e)
                if (gr + e < M) out[(int64_t)gc * ld_out + gr + e] = tmp[e];
        }
    }
}

// RMSNorm backward.  Each workgroup walks `rows_per_block` rows; a thread owns fixed columns so the
// weight gradient is accumulated in registers and flushed with one atomicAdd per column per workgroup.
template <int VPT>
__global__ __launch_bounds__(NT) void rmsnorm_bwd_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                         const uint16_t* __restrict__ w, const uint16_t* __restrict__ dres,
                                                         uint16_t* __restrict__ dx, float* __restrict__ dw,
                                                         float* __restrict__ ws, int M, int h, float eps, int rows_per_block) {
    __shared__ float red[NT / 64];
    const int nv = h >> 3;
    float wv[VPT][8], dwacc[VPT][8];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dwacc[i][e] = 0.f; wv[i][e] = 0.f; }
        if (v < nv) unpack8(*(const u32x4*)(w + v * 8), wv[i]);
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int row = r0; row < r1; ++row) {
        const uint16_t* xr = x + (int64_t)row * h;
        const uint16_t* gr = dy + (int64_t)row * h;
        float xv[VPT][8], gv[VPT][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
                unpack8(*(const u32x4*)(xr + v * 8), xv[i]);
                unpack8(*(const u32x4*)(gr + v * 8), gv[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += xv[i][e] * xv[i][e];
            }
        }
        ss = block_sum<NT>(ss, red);
        const float rstd = rsqrtf(ss / (float)h + eps);
        float dot = 0.f;                                    // sum_c (dy*w) * xhat
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = xv[i][e] * rstd;
                    dwacc[i][e] += gv[i][e] * round_bf(xh);
                    gv[i][e] *= wv[i][e];
                    dot += gv[i][e] * xh;
                }
            }
        }
        dot = block_sum<NT>(dot, red) / (float)h;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
                float o[8];
                if (dres) unpack8(*(const u32x4*)(dres + (int64_t)row * h + v * 8), o);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += rstd * (gv[i][e] - xv[i][e] * rstd * dot);
                *(u32x4*)(dx + (int64_t)row * h + v * 8) = pack8(o);
            }
        }
    }
    if (ws) {                                                // per-workgroup partial row: summed by rmsnorm_dw_reduce_kernel
        float* wr = ws + (int64_t)blockIdx.x * h;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
                *(f32x4*)(wr + v * 8) = f32x4{dwacc[i][0], dwacc[i][1], dwacc[i][2], dwacc[i][3]};
                *(f32x4*)(wr + v * 8 + 4) = f32x4{dwacc[i][4], dwacc[i][5], dwacc[i][6], dwacc[i][7]};
            }
        }
    } else if (dw) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(dw + v * 8 + e, dwacc[i][e]);
            }
        }
    }
}

// LayerNorm forward (fp32 math, one rounding at the end, like torch.layer_norm on bf16).
template <int VPT>
__global__ __launch_bounds__(NT) void layernorm_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                           const uint16_t* __restrict__ b, uint16_t* __restrict__ y, int h, float eps) {
    __shared__ float red[NT / 64];
    const int64_t row = blockIdx.x;
    const int nv = h >> 3;
    float xv[VPT][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nv) {
            unpack8(*(const u32x4*)(x + row * h + v * 8), xv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xv[i][e];
        }
    }
    const float mean = block_sum<NT>(s, red) / (float)h;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = xv[i][e] - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(block_sum<NT>(ss, red) / (float)h + eps);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nv) {
            float wv[8], bv[8], o[8];
            unpack8(*(const u32x4*)(w + v * 8), wv);
            unpack8(*(const u32x4*)(b + v * 8), bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (xv[i][e] - mean) * rstd * wv[e] + bv[e];
            *(u32x4*)(y + row * h + v * 8) = pack8(o);
        }
    }
}

// LayerNorm backward (trainable SigLIP tower).  Same walk as rmsnorm_bwd_kernel: a workgroup handles
// `rows_per_block` rows, a thread owns fixed columns and keeps their dw / db partial sums in registers; the partial rows go to
// the caller's workspace ([workgroup][2h]: dw | db) and are summed in a fixed order by rmsnorm_dw_reduce_kernel (no atomics).
//   g = dy * w;  dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat));  dw += dy * xhat;  db += dy
template <int VPT>
__global__ __launch_bounds__(NT) void layernorm_bwd_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                           const uint16_t* __restrict__ w, const uint16_t* __restrict__ dres,
                                                           uint16_t* __restrict__ dx, float* __restrict__ ws, int M, int h, float eps,
                                                           int rows_per_block) {
    __shared__ float red[NT / 64];
    const int nv = h >> 3;
    float wv[VPT][8], dwacc[VPT][8], dbacc[VPT][8];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dwacc[i][e] = 0.f; dbacc[i][e] = 0.f; wv[i][e] = 0.f; }
        if (v < nv) unpack8(*(const u32x4*)(w + v * 8), wv[i]);
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int row = r0; row < r1; ++row) {
        float xv[VPT][8], gv[VPT][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
                unpack8(*(const u32x4*)(x + (int64_t)row * h + v * 8), xv[i]);
                unpack8(*(const u32x4*)(dy + (int64_t)row * h + v * 8), gv[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += xv[i][e];
            }
        }
        const float mean = block_sum<NT>(s, red) / (float)h;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[i][e] -= mean; ss += xv[i][e] * xv[i][e]; }
            }
        }
        const float rstd = rsqrtf(block_sum<NT>(ss, red) / (float)h + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = xv[i][e] * rstd;
                    xv[i][e] = xh;
                    dwacc[i][e] += gv[i][e] * xh;
                    dbacc[i][e] += gv[i][e];
                    gv[i][e] *= wv[i][e];
                    sg += gv[i][e];
                    sgx += gv[i][e] * xh;
                }
            }
        }
        sg = block_sum<NT>(sg, red) / (float)h;
        sgx = block_sum<NT>(sgx, red) / (float)h;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NT;
            if (v < nv) {
                float o[8];
                if (dres) unpack8(*(const u32x4*)(dres + (int64_t)row * h + v * 8), o);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += rstd * (gv[i][e] - sg - xv[i][e] * sgx);
                *(u32x4*)(dx + (int64_t)row * h + v * 8) = pack8(o);
            }
        }
    }
    float* wr = ws + (int64_t)blockIdx.x * 2 * h;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nv) {
            *(f32x4*)(wr + v * 8) = f32x4{dwacc[i][0], dwacc[i][1], dwacc[i][2], dwacc[i][3]};
            *(f32x4*)(wr + v * 8 + 4) = f32x4{dwacc[i][4], dwacc[i][5], dwacc[i][6], dwacc[i][7]};
            *(f32x4*)(wr + h + v * 8) = f32x4{dbacc[i][0], dbacc[i][1], dbacc[i][2], dbacc[i][3]};
            *(f32x4*)(wr + h + v * 8 + 4) = f32x4{dbacc[i][4], dbacc[i][5], dbacc[i][6], dbacc[i][7]};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Cross entropy over rows of bf16 logits, gradient written in place.   one workgroup (512 thr) per row.
// pass 1: online max / sum-exp;  pass 2: g = scale * (softmax - onehot)
// ------------------------------------------------------------------------------------------------
constexpr int CE_NT = 512;
__global__ __launch_bounds__(CE_NT) void ce_rows_kernel(uint16_t* __restrict__ logits, int64_t ld, const int32_t* __restrict__ targets,
                                                        int V, float grad_scale, float* __restrict__ loss_sum, float* __restrict__ row_out) {
    __shared__ float red[CE_NT / 64];
    const int64_t row = blockIdx.x;
    uint16_t* lr = logits + row * ld;
    const int tgt = targets[row];
    const int nvec = (int)(ld >> 3);
    if (tgt < 0) {                                           // ignored row: zero gradient
        for (int v = threadIdx.x; v < nvec; v += CE_NT) *(u32x4*)(lr + v * 8) = u32x4{0u, 0u, 0u, 0u};
        if (row_out && threadIdx.x == 0) row_out[row] = 0.f;
        return;
    }
    float m = -INFINITY, s = 0.f;
    for (int v = threadIdx.x; v < nvec; v += CE_NT) {
        float f[8];
        unpack8(*(const u32x4*)(lr + v * 8), f);
        float lm = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (v * 8 + e >= V) f[e] = -INFINITY; lm = fmaxf(lm, f[e]); }
        if (lm > m) { s *= __expf(m - lm); m = lm; }
        if (m > -INFINITY) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s += __expf(f[e] - m);
        }
    }
    const float gmax = block_max<CE_NT>(m, red);
    s = (m > -INFINITY) ? s * __expf(m - gmax) : 0.f;
    const float gsum = block_sum<CE_NT>(s, red);
    const float lse = gmax + __logf(gsum);
    if (threadIdx.x == 0) {
        if (row_out) row_out[row] = lse - bf2f(lr[tgt]);    // summed in a fixed order afterwards (mm_sum_rows_kernel)
        else atomicAdd(loss_sum, lse - bf2f(lr[tgt]));
    }
    __syncthreads();                                         // target logit read before it is overwritten
    for (int v = threadIdx.x; v < nvec; v += CE_NT) {
        float f[8];
        unpack8(*(const u32x4*)(lr + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = v * 8 + e;
            float g = (c < V) ? __expf(f[e] - lse) : 0.f;
            if (c == tgt) g -= 1.0f;
            f[e] = g * grad_scale;
        }
        *(u32x4*)(lr + v * 8) = pack8(f);
    }
}

// ------------------------------------------------------------------------------------------------
// Cosine regression loss, one wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void cosine_loss_kernel(const uint16_t* __restrict__ pred, const uint16_t* __restrict__ tgt, int R, int C,
                                                         int normalize, float* __restrict__ cos_sum, uint16_t* __restrict__ dpred,
                                                         float* __restrict__ row_out) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int nv = C >> 3;
    const uint16_t* pr = pred + (int64_t)row * C;
    const uint16_t* tr = tgt + (int64_t)row * C;
    float pp = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float p[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
#pragma unroll
        for (int e = 0; e < 8; ++e) pp += p[e] * p[e];
    }
    pp = wave_sum(pp);
    // F.normalize on a bf16 tensor: norm rounded to bf16, clamp, divide, round
    const float pn = normalize ? fmaxf(round_bf(sqrtf(pp)), 1e-12f) : 1.0f;
    float tt = 0.f, uu = 0.f, tu = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float p[8], t[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
        unpack8(*(const u32x4*)(tr + v * 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = normalize ? round_bf(p[e] / pn) : p[e];
            tt += t[e] * t[e]; uu += u * u; tu += t[e] * u;
        }
    }
    tt = wave_sum(tt); uu = wave_sum(uu); tu = wave_sum(tu);
    const float nt = fmaxf(sqrtf(tt), 1e-8f), nu = fmaxf(sqrtf(uu), 1e-8f);
    const float c = tu / (nt * nu);
    if (lane == 0) {
        if (row_out) row_out[row] = c;
        else atomicAdd(cos_sum, c);
    }
    if (!dpred) return;
    // w = d cos / d u = (t/nt - c * u/nu) / nu ;  d u / d p = (I - phat phat^T) / pn
    const float inv_r = -1.0f / (float)R;
    float pw = 0.f;                                          // phat . w  (only when normalising)
    const float pnorm = sqrtf(pp);
    if (normalize) {
        for (int v = lane; v < nv; v += 64) {
            float p[8], t[8];
            unpack8(*(const u32x4*)(pr + v * 8), p);
            unpack8(*(const u32x4*)(tr + v * 8), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = round_bf(p[e] / pn);
                const float w = (t[e] / nt - c * u / nu) / nu;
                pw += p[e] * w;
            }
        }
        pw = wave_sum(pw) / fmaxf(pnorm, 1e-20f);
    }
    for (int v = lane; v < nv; v += 64) {
        float p[8], t[8], g[8];
        unpack8(*(const u32x4*)(pr + v * 8), p);
        unpack8(*(const u32x4*)(tr + v * 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = normalize ? round_bf(p[e] / pn) : p[e];
            const float w = (t[e] / nt - c * u / nu) / nu;
            g[e] = normalize ? inv_r * (w - (p[e] / fmaxf(pnorm, 1e-20f)) * pw) / pn : inv_r * w;
        }
        *(u32x4*)(dpred + (int64_t)row * C + v * 8) = pack8(g);
    }
}

// ------------------------------------------------------------------------------------------------
// Bilinear token reduction (fp32, align_corners=False) + L2 normalise; one wave per output token.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lerp_src(int o, int in_size, int out_size, int& i0, int& i1, float& w1) {
    const float scale = (float)in_size / (float)out_size;
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    w1 = src - (float)i0;
}

__global__ __launch_bounds__(NT) void bilinear_l2norm_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int N, int si, int so,
                                                             int C, int normalize) {
    const int tok = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= N * so * so) return;
    const int n = tok / (so * so), oy = (tok / so) % so, ox = tok % so;
    const int nv = C >> 3;
    int y0, y1, x0, x1; float ly, lx;
    if (si == so) { y0 = y1 = oy; x0 = x1 = ox; ly = lx = 0.f; }
    else { lerp_src(oy, si, so, y0, y1, ly); lerp_src(ox, si, so, x0, x1, lx); }
    const uint16_t* b = in + (int64_t)n * si * si * C;
    const uint16_t* p00 = b + (int64_t)(y0 * si + x0) * C;
    const uint16_t* p01 = b + (int64_t)(y0 * si + x1) * C;
    const uint16_t* p10 = b + (int64_t)(y1 * si + x0) * C;
    const uint16_t* p11 = b + (int64_t)(y1 * si + x1) * C;
    uint16_t* o = out + (int64_t)tok * C;
    float ss = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float a[8], bq[8], c[8], d[8], r[8];
        unpack8(*(const u32x4*)(p00 + v * 8), a);
        unpack8(*(const u32x4*)(p01 + v * 8), bq);
        unpack8(*(const u32x4*)(p10 + v * 8), c);
        unpack8(*(const u32x4*)(p11 + v * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float top = a[e] * (1.0f - lx) + bq[e] * lx;
            const float bot = c[e] * (1.0f - lx) + d[e] * lx;
            r[e] = round_bf(si == so ? a[e] : top * (1.0f - ly) + bot * ly);
            ss += r[e] * r[e];
        }
        *(u32x4*)(o + v * 8) = pack8(r);
    }
    if (!normalize) return;
    ss = wave_sum(ss);
    const float nrm = fmaxf(round_bf(sqrtf(ss)), 1e-12f);
    for (int v = lane; v < nv; v += 64) {                    // same lane re-reads what it wrote
        float r[8];
        unpack8(*(const u32x4*)(o + v * 8), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = r[e] / nrm;
        *(u32x4*)(o + v * 8) = pack8(r);
    }
}

// Backward of bilinear_l2norm_kernel (trainable tower): one wave per OUTPUT token recomputes the interpolated
// row r and its norm, forms dr = (dy - y (y . dy)) / |r| (or dy when not normalising) and scatters it onto the four source
// pixels with the bilinear weights (fp32 atomics into a caller-zeroed buffer: a source pixel feeds a handful of outputs).
__global__ __launch_bounds__(NT) void bilinear_l2norm_bwd_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ dy,
                                                                 float* __restrict__ din, int N, int si, int so, int C, int normalize) {
    const int tok = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= N * so * so) return;
    const int n = tok / (so * so), oy = (tok / so) % so, ox = tok % so;
    const int nv = C >> 3;
    int y0, y1, x0, x1; float ly, lx;
    if (si == so) { y0 = y1 = oy; x0 = x1 = ox; ly = lx = 0.f; }
    else { lerp_src(oy, si, so, y0, y1, ly); lerp_src(ox, si, so, x0, x1, lx); }
    const int64_t base = (int64_t)n * si * si * C;
    const int64_t o00 = base + (int64_t)(y0 * si + x0) * C, o01 = base + (int64_t)(y0 * si + x1) * C;
    const int64_t o10 = base + (int64_t)(y1 * si + x0) * C, o11 = base + (int64_t)(y1 * si + x1) * C;
    const uint16_t* g = dy + (int64_t)tok * C;
    auto interp = [&](int v, float* r) {
        float a[8], bq[8], c[8], d[8];
        unpack8(*(const u32x4*)(in + o00 + v * 8), a);
        unpack8(*(const u32x4*)(in + o01 + v * 8), bq);
        unpack8(*(const u32x4*)(in + o10 + v * 8), c);
        unpack8(*(const u32x4*)(in + o11 + v * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float top = a[e] * (1.0f - lx) + bq[e] * lx;
            const float bot = c[e] * (1.0f - lx) + d[e] * lx;
            r[e] = round_bf(si == so ? a[e] : top * (1.0f - ly) + bot * ly);
        }
    };
    float inv = 1.0f, dot = 0.f;
    if (normalize) {
        float ss = 0.f;
        for (int v = lane; v < nv; v += 64) {
            float r[8], gy[8];
            interp(v, r);
            unpack8(*(const u32x4*)(g + v * 8), gy);
#pragma unroll
            for (int e = 0; e < 8; ++e) { ss += r[e] * r[e]; dot += r[e] * gy[e]; }
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float nrm = fmaxf(round_bf(sqrtf(ss)), 1e-12f);
        inv = 1.0f / nrm;
        dot *= inv * inv;                                    // (y . dy) / |r|  with y = r / |r|
    }
    const float w00 = (1.0f - lx) * (1.0f - ly), w01 = lx * (1.0f - ly), w10 = (1.0f - lx) * ly, w11 = lx * ly;
    for (int v = lane; v < nv; v += 64) {
        float r[8], gy[8];
        unpack8(*(const u32x4*)(g + v * 8), gy);
        if (normalize) interp(v, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dr = normalize ? (gy[e] - r[e] * dot) * inv : gy[e];
            if (si == so) atomicAdd(din + o00 + v * 8 + e, dr);
            else {
                atomicAdd(din + o00 + v * 8 + e, dr * w00);
                atomicAdd(din + o01 + v * 8 + e, dr * w01);
                atomicAdd(din + o10 + v * 8 + e, dr * w10);
                atomicAdd(din + o11 + v * 8 + e, dr * w11);
            }
        }
    }
}

template <typename F>
int dispatch_vpt(int h, F&& f) {
    const int nv = h >> 3;
    if (nv <= NT) return f(std::integral_constant<int, 1>{});
    if (nv <= 2 * NT) return f(std::integral_constant<int, 2>{});
    if (nv <= 4 * NT) return f(std::integral_constant<int, 4>{});
    if (nv <= 8 * NT) return f(std::integral_constant<int, 8>{});
    return MM355_EUNSUPPORTED;
}

}  // namespace

extern "C" int mm355_rmsnorm_fwd_rstd(const mm355_bf16* x, const mm355_bf16* w, mm355_bf16* y, float* rstd_out, int64_t M, int64_t h, float eps,
                                      void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !w || !y || M <= 0 || h <= 0 || (h & 7) || M > 0x7fffffff) return MM355_EINVAL;
    return dispatch_vpt((int)h, [&](auto vpt) {
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<decltype(vpt)::value>), dim3((unsigned)M), dim3(NT), 0, (hipStream_t)stream, x, w, y, rstd_out,
                           (int)h, eps);
        return mm_launch_status();
    });
}
extern "C" int mm355_rmsnorm_fwd(const mm355_bf16* x, const mm355_bf16* w, mm355_bf16* y, int64_t M, int64_t h, float eps, void* stream) {
    return mm355_rmsnorm_fwd_rstd(x, w, y, nullptr, M, h, eps, stream);
}
extern "C" int mm355_rmsnorm_apply_t(const mm355_bf16* x, const mm355_bf16* w, const float* rstd, int64_t M, int64_t h, mm355_bf16* out_t,
                                     int64_t ld_out, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !w || !rstd || !out_t || M <= 0 || h <= 0 || (h & 7) || M > 0x7fffffff || ld_out < M) return MM355_EINVAL;
    if (!mm_aligned16(x) || !mm_aligned16(w) || !mm_aligned16(out_t)) return MM355_EINVAL;
    dim3 grid((unsigned)((h + 63) / 64), (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL(rmsnorm_apply_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, rstd, (int)M, (int)h, out_t, ld_out);
    return mm_launch_status();
}

// dw[c] += sum_g ws[g][c]: 32 columns per workgroup as eight float4 lanes x 32 row lanes (16-B loads, G / 32 of them per
// thread), then a fixed-order sum over the row lanes (deterministic).  h % 4 == 0, ws rows 16-B aligned.
constexpr int DWR_COLS = 32;
__global__ __launch_bounds__(256) void rmsnorm_dw_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int G, int h, int ld) {
    __shared__ f32x4 part[32][8];
    const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * DWR_COLS + cq * 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (c < h) {
        const float* p = ws + c;
        int g = rl;
        for (; g + 32 < G; g += 64) {                         // two independent chains
            acc0 += *(const f32x4*)(p + (int64_t)g * ld);
            acc1 += *(const f32x4*)(p + (int64_t)(g + 32) * ld);
        }
        if (g < G) acc0 += *(const f32x4*)(p + (int64_t)g * ld);
    }
    part[rl][cq] = acc0 + acc1;
    __syncthreads();
    if (rl == 0 && c < h) {
        f32x4 s = part[0][cq];
#pragma unroll
        for (int r = 1; r < 32; ++r) s += part[r][cq];
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[c + e] += s[e];
    }
}

// The same fixed-order sum, landing straight in a bf16 gradient buffer: grad = (accumulate ? grad : 0) + sum  (one rounding).
__global__ __launch_bounds__(256) void rmsnorm_dw_reduce_bf16_kernel(const float* __restrict__ ws, uint16_t* __restrict__ grad, int G, int h, int ld,
                                                                     int accumulate) {
    __shared__ f32x4 part[32][8];
    const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * DWR_COLS + cq * 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (c < h) {
        const float* p = ws + c;
        int g = rl;
        for (; g + 32 < G; g += 64) {
            acc0 += *(const f32x4*)(p + (int64_t)g * ld);
            acc1 += *(const f32x4*)(p + (int64_t)(g + 32) * ld);
        }
        if (g < G) acc0 += *(const f32x4*)(p + (int64_t)g * ld);
    }
    part[rl][cq] = acc0 + acc1;
    __syncthreads();
    if (rl == 0 && c < h) {
        f32x4 s = part[0][cq];
#pragma unroll
        for (int r = 1; r < 32; ++r) s += part[r][cq];
#pragma unroll
        for (int e = 0; e < 4; ++e) grad[c + e] = f2bf(s[e] + (accumulate ? bf2f(grad[c + e]) : 0.f));
    }
}

static int rmsnorm_bwd_rows_per_block(int64_t M, bool two_stage) {
    if (two_stage) return M >= 8192 ? 32 : (M >= 1024 ? 8 : 1);
    return M >= 8192 ? 16 : (M >= 1024 ? 4 : 1);
}

extern "C" int64_t mm355_rmsnorm_bwd_ws_floats(int64_t M, int64_t h) {
    if (M <= 0 || h <= 0) return 0;
    const int rpb = rmsnorm_bwd_rows_per_block(M, true);
    return ((M + rpb - 1) / rpb) * h;
}

extern "C" int mm355_rmsnorm_bwd(const mm355_bf16* dy, const mm355_bf16* x, const mm355_bf16* w, const mm355_bf16* dres,
                                 mm355_bf16* dx, float* dw_f32, float* workspace, int64_t M, int64_t h, float eps, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!dy || !x || !w || !dx || M <= 0 || h <= 0 || (h & 7) || M > 0x7fffffff) return MM355_EINVAL;
    const bool two_stage = dw_f32 && workspace;
    if (two_stage && !mm_aligned16(workspace)) return MM355_EINVAL;
    const int rpb = rmsnorm_bwd_rows_per_block(M, two_stage);
    const unsigned grid = (unsigned)((M + rpb - 1) / rpb);
    int rc = dispatch_vpt((int)h, [&](auto vpt) {
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<decltype(vpt)::value>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, dy, x, w, dres, dx,
                           dw_f32, two_stage ? workspace : nullptr, (int)M, (int)h, eps, rpb);
        return mm_launch_status();
    });
    if (rc != MM355_OK || !two_stage) return rc;
    hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel, dim3((unsigned)((h + DWR_COLS - 1) / DWR_COLS)), dim3(256), 0, (hipStream_t)stream, workspace, dw_f32,
                       (int)grid, (int)h, (int)h);
    return mm_launch_status();
}

extern "C" int mm355_rmsnorm_bwd_wgrad(const mm355_bf16* dy, const mm355_bf16* x, const mm355_bf16* w, const mm355_bf16* dres, mm355_bf16* dx,
                                       mm355_bf16* w_grad, int accumulate, float* workspace, int64_t M, int64_t h, float eps, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!dy || !x || !w || !dx || !w_grad || !workspace || M <= 0 || h <= 0 || (h & 7) || M > 0x7fffffff || !mm_aligned16(workspace)) return MM355_EINVAL;
    const int rpb = rmsnorm_bwd_rows_per_block(M, true);
    const unsigned grid = (unsigned)((M + rpb - 1) / rpb);
    int rc = dispatch_vpt((int)h, [&](auto vpt) {
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<decltype(vpt)::value>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, dy, x, w, dres, dx,
                           (float*)nullptr, workspace, (int)M, (int)h, eps, rpb);
        return mm_launch_status();
    });
    if (rc != MM355_OK) return rc;
    hipLaunchKernelGGL(rmsnorm_dw_reduce_bf16_kernel, dim3((unsigned)((h + DWR_COLS - 1) / DWR_COLS)), dim3(256), 0, (hipStream_t)stream, workspace,
                       w_grad, (int)grid, (int)h, (int)h, accumulate);
    return mm_launch_status();
}

extern "C" int mm355_layernorm_fwd(const mm355_bf16* x, const mm355_bf16* w, const mm355_bf16* b, mm355_bf16* y, int64_t M, int64_t h,
                                   float eps, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !w || !b || !y || M <= 0 || h <= 0 || (h & 7) || M > 0x7fffffff) return MM355_EINVAL;
    return dispatch_vpt((int)h, [&](auto vpt) {
        hipLaunchKernelGGL((layernorm_fwd_kernel<decltype(vpt)::value>), dim3((unsigned)M), dim3(NT), 0, (hipStream_t)stream, x, w, b, y, (int)h, eps);
        return mm_launch_status();
    });
}

extern "C" int64_t mm355_layernorm_bwd_ws_floats(int64_t M, int64_t h) {
    if (M <= 0 || h <= 0) return 0;
    const int rpb = rmsnorm_bwd_rows_per_block(M, true);
    return ((M + rpb - 1) / rpb) * 2 * h;
}

extern "C" int mm355_layernorm_bwd(const mm355_bf16* dy, const mm355_bf16* x, const mm355_bf16* w, const mm355_bf16* dres, mm355_bf16* dx,
                                   float* dw_f32, float* db_f32, float* workspace, int64_t M, int64_t h, float eps, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!dy || !x || !w || !dx || !dw_f32 || !db_f32 || !workspace || M <= 0 || h <= 0 || (h & 7) || M > 0x7fffffff || !mm_aligned16(workspace))
        return MM355_EINVAL;
    const int rpb = rmsnorm_bwd_rows_per_block(M, true);
    const unsigned grid = (unsigned)((M + rpb - 1) / rpb);
    int rc = dispatch_vpt((int)h, [&](auto vpt) {
        hipLaunchKernelGGL((layernorm_bwd_kernel<decltype(vpt)::value>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, dy, x, w, dres, dx,
                           workspace, (int)M, (int)h, eps, rpb);
        return mm_launch_status();
    });
    if (rc != MM355_OK) return rc;
    const dim3 rg((unsigned)((h + DWR_COLS - 1) / DWR_COLS));
    hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel, rg, dim3(256), 0, (hipStream_t)stream, workspace, dw_f32, (int)grid, (int)h, (int)(2 * h));
    hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel, rg, dim3(256), 0, (hipStream_t)stream, workspace + h, db_f32, (int)grid, (int)h, (int)(2 * h));
    return mm_launch_status();
}

extern "C" int mm355_sum_rows_f32(const float* values, int64_t n, float scale, float* out, int accumulate, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!values || !out || n <= 0) return MM355_EINVAL;
    hipLaunchKernelGGL(mm_sum_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, values, n, scale, out, accumulate);
    return mm_launch_status();
}

extern "C" int mm355_ce_rows(mm355_bf16* logits, int64_t ld, const int32_t* targets, int64_t R, int64_t V, float grad_scale,
                             float* loss_sum, float* row_ws, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!logits || !targets || (!loss_sum && !row_ws) || R <= 0 || V <= 0 || ld < V || (ld & 7) || R > 0x7fffffff || !mm_aligned16(logits))
        return MM355_EINVAL;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)R), dim3(CE_NT), 0, (hipStream_t)stream, logits, ld, targets, (int)V, grad_scale, loss_sum,
                       row_ws);
    if (row_ws && loss_sum) hipLaunchKernelGGL(mm_sum_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)row_ws, R, 1.0f, loss_sum, 1);
    return mm_launch_status();
}

extern "C" int mm355_cosine_loss(const mm355_bf16* pred_raw, const mm355_bf16* target, int64_t R, int64_t C, int normalize,
                                 float* cos_sum, mm355_bf16* dpred, float* row_ws, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!pred_raw || !target || !cos_sum || R <= 0 || C <= 0 || (C & 7)) return MM355_EINVAL;
    const unsigned grid = (unsigned)((R + NT / 64 - 1) / (NT / 64));
    hipLaunchKernelGGL(cosine_loss_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, pred_raw, target, (int)R, (int)C, normalize, cos_sum, dpred,
                       row_ws);
    if (row_ws) hipLaunchKernelGGL(mm_sum_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)row_ws, R, 1.0f, cos_sum, 1);
    return mm_launch_status();
}

extern "C" int mm355_bilinear_l2norm_bwd(const mm355_bf16* in, const mm355_bf16* d_out, float* d_in_f32, int64_t N, int64_t side_in, int64_t side_out,
                                         int64_t C, int normalize, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!in || !d_out || !d_in_f32 || N <= 0 || side_in <= 0 || side_out <= 0 || C <= 0 || (C & 7)) return MM355_EINVAL;
    const int64_t toks = N * side_out * side_out;
    const unsigned grid = (unsigned)((toks + NT / 64 - 1) / (NT / 64));
    hipLaunchKernelGGL(bilinear_l2norm_bwd_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, in, d_out, d_in_f32, (int)N, (int)side_in,
                       (int)side_out, (int)C, normalize);
    return mm_launch_status();
}

extern "C" int mm355_bilinear_l2norm(const mm355_bf16* in, mm355_bf16* out, int64_t N, int64_t side_in, int64_t side_out, int64_t C,
                                     int normalize, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!in || !out || N <= 0 || side_in <= 0 || side_out <= 0 || C <= 0 || (C & 7)) return MM355_EINVAL;
    const int64_t toks = N * side_out * side_out;
    const unsigned grid = (unsigned)((toks + NT / 64 - 1) / (NT / 64));
    hipLaunchKernelGGL(bilinear_l2norm_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, in, out, (int)N, (int)side_in, (int)side_out, (int)C, normalize);
    return mm_launch_status();
}
