// Elementwise / gather-scatter HBM-bound kernels (gfx950): RoPE, SwiGLU, GELU, casts, AdamW shard update,
// grad-norm, per-head transposes, im2col for the SigLIP patch embedding, splice gather/scatter.
#include "mm355_common.h"
#include <algorithm>

namespace {

constexpr int NT = 256;
constexpr int MAX_GRID = 256 * 16;      // grid-stride cap: 16 workgroups per CU

inline unsigned grid_for(int64_t work_items) {
    int64_t g = (work_items + NT - 1) / NT;
    if (g > MAX_GRID) g = MAX_GRID;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------------ RoPE
__global__ void rope_table_kernel(uint16_t* __restrict__ cos_o, uint16_t* __restrict__ sin_o, int L, int d, float theta) {
    const int half = d >> 1;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < (int64_t)L * half; i += (int64_t)gridDim.x * NT) {
        const int l = (int)(i / half), j = (int)(i % half);
        const float inv = 1.0f / powf(theta, (float)(2 * j) / (float)d);
        const float ang = (float)l * inv;
        const uint16_t c = f2bf(cosf(ang)), s = f2bf(sinf(ang));
        cos_o[(int64_t)l * d + j] = c; cos_o[(int64_t)l * d + half + j] = c;
        sin_o[(int64_t)l * d + j] = s; sin_o[(int64_t)l * d + half + j] = s;
    }
}

// The same tables from per-band inverse frequencies the HOST computed (scaled RoPE variants: HF ROPE_INIT_FUNCTIONS rescale inv_freq per
// wavelength band once at construction; LlamaRotaryEmbedding.forward is then cos / sin(position * inv_freq) * attention_scaling -> bf16).
__global__ void rope_table_freq_kernel(uint16_t* __restrict__ cos_o, uint16_t* __restrict__ sin_o, int L, int d,
                                       const float* __restrict__ inv_freq, float scaling) {
    const int half = d >> 1;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < (int64_t)L * half; i += (int64_t)gridDim.x * NT) {
        const int l = (int)(i / half), j = (int)(i % half);
        const float ang = (float)l * inv_freq[j];
        const uint16_t c = f2bf(cosf(ang) * scaling), s = f2bf(sinf(ang) * scaling);
        cos_o[(int64_t)l * d + j] = c; cos_o[(int64_t)l * d + half + j] = c;
        sin_o[(int64_t)l * d + j] = s; sin_o[(int64_t)l * d + half + j] = s;
    }
}

// In-place rotation of the q and k column blocks of a fused qkv activation.  One thread handles 8
// consecutive elements of the first half of a head together with their partners in the second half.
// forward : y1 = bf(bf(x1*c) + bf(-x2*s)),  y2 = bf(bf(x2*c) + bf(x1*s))     (HF rounding order)
// inverse : dx1 = dy1*c + dy2*s,            dx2 = dy2*c - dy1*s
__global__ __launch_bounds__(NT) void rope_qk_kernel(uint16_t* __restrict__ qkv, int64_t ld, int B, int L, int H, int d,
                                                     const uint16_t* __restrict__ cos_t, const uint16_t* __restrict__ sin_t, int inverse,
                                                     const int32_t* __restrict__ pos_off) {
    const int half = d >> 1, vph = half >> 3;               // vectors per half head
    const int64_t total = (int64_t)B * L * H * vph;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int v = (int)(i % vph);
        const int hd = (int)((i / vph) % H);
        const int64_t row = i / ((int64_t)vph * H);
        const int l = (int)(row % L) + (pos_off ? pos_off[row / L] : 0);     // table row = position id (left padding: row index + offset)
        uint16_t* p1 = qkv + row * ld + (int64_t)hd * d + v * 8;
        uint16_t* p2 = p1 + half;
        float x1[8], x2[8], c[8], s[8], y1[8], y2[8];
        unpack8(*(const u32x4*)p1, x1);
        unpack8(*(const u32x4*)p2, x2);
        unpack8(*(const u32x4*)(cos_t + (int64_t)l * d + v * 8), c);
        unpack8(*(const u32x4*)(sin_t + (int64_t)l * d + v * 8), s);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!inverse) {
                y1[e] = round_bf(x1[e] * c[e]) + round_bf(-x2[e] * s[e]);
                y2[e] = round_bf(x2[e] * c[e]) + round_bf(x1[e] * s[e]);
            } else {
                y1[e] = x1[e] * c[e] + x2[e] * s[e];
                y2[e] = x2[e] * c[e] - x1[e] * s[e];
            }
        }
        *(u32x4*)p1 = pack8(y1);
        *(u32x4*)p2 = pack8(y2);
    }
}

// ------------------------------------------------------------------------------------------------ SwiGLU / GELU
__global__ __launch_bounds__(NT) void swiglu_fwd_kernel(const uint16_t* __restrict__ gu, uint16_t* __restrict__ act, int64_t M, int I) {
    const int iv = I >> 3;
    const int64_t total = M * iv;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t row = i / iv; const int v = (int)(i % iv);
        float g[8], u[8], o[8];
        unpack8(*(const u32x4*)(gu + row * 2 * I + v * 8), g);
        unpack8(*(const u32x4*)(gu + row * 2 * I + I + v * 8), u);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = round_bf(g[e] / (1.0f + __expf(-g[e]))) * u[e];
        *(u32x4*)(act + row * I + v * 8) = pack8(o);
    }
}
__global__ __launch_bounds__(NT) void swiglu_bwd_kernel(const uint16_t* __restrict__ gu, const uint16_t* __restrict__ dact, uint16_t* __restrict__ dgu,
                                                        uint16_t* __restrict__ act, int64_t M, int I) {
    const int iv = I >> 3;
    const int64_t total = M * iv;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t row = i / iv; const int v = (int)(i % iv);
        float g[8], u[8], da[8], dg[8], du[8], a[8];
        unpack8(*(const u32x4*)(gu + row * 2 * I + v * 8), g);
        unpack8(*(const u32x4*)(gu + row * 2 * I + I + v * 8), u);
        unpack8(*(const u32x4*)(dact + row * I + v * 8), da);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = 1.0f / (1.0f + __expf(-g[e]));
            const float silu = g[e] * sg;
            a[e] = round_bf(silu) * u[e];
            du[e] = da[e] * round_bf(silu);
            dg[e] = da[e] * u[e] * (sg * (1.0f + g[e] * (1.0f - sg)));
        }
        *(u32x4*)(dgu + row * 2 * I + v * 8) = pack8(dg);
        *(u32x4*)(dgu + row * 2 * I + I + v * 8) = pack8(du);
        if (act) *(u32x4*)(act + row * I + v * 8) = pack8(a);
    }
}
// SwiGLU backward with the contraction-major copies the two weight-gradient GEMMs want, written from the same registers:
// 64 x 64 tiles; dgu goes out row-major as usual, act / dgate / dup additionally through three LDS tiles as
// actT[I][M], dguT[2I][M] (what mm355_transpose_bf16 would produce from act and dgu, without re-reading them).
__global__ __launch_bounds__(256) void swiglu_bwd_t_kernel(const uint16_t* __restrict__ gu, const uint16_t* __restrict__ dact,
                                                           uint16_t* __restrict__ dgu, uint16_t* __restrict__ actT,
                                                           uint16_t* __restrict__ dguT, int64_t M, int I) {
    __shared__ uint16_t tile[3][64][64 + 2];
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256, r = v >> 3, c = (v & 7) * 8;
        const int64_t row = r0 + r;
        float g[8], u[8], da[8], dg[8], du[8], a[8];
        unpack8(*(const u32x4*)(gu + row * 2 * I + c0 + c), g);
        unpack8(*(const u32x4*)(gu + row * 2 * I + I + c0 + c), u);
        unpack8(*(const u32x4*)(dact + row * I + c0 + c), da);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = 1.0f / (1.0f + __expf(-g[e]));
            const float silu = g[e] * sg;
            a[e] = round_bf(silu) * u[e];
            du[e] = da[e] * round_bf(silu);
            dg[e] = da[e] * u[e] * (sg * (1.0f + g[e] * (1.0f - sg)));
        }
        const u32x4 pg = pack8(dg), pu = pack8(du), pa = pack8(a);
        *(u32x4*)(dgu + row * 2 * I + c0 + c) = pg;
        *(u32x4*)(dgu + row * 2 * I + I + c0 + c) = pu;
        const uint16_t* sa = (const uint16_t*)&pa; const uint16_t* sg16 = (const uint16_t*)&pg; const uint16_t* su = (const uint16_t*)&pu;
#pragma unroll
        for (int e = 0; e < 8; ++e) { tile[0][r][c + e] = sa[e]; tile[1][r][c + e] = sg16[e]; tile[2][r][c + e] = su[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256, c = v >> 3, r = (v & 7) * 8;   // output row = input column c
        uint16_t ta[8], tg[8], tu[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ta[e] = tile[0][r + e][c]; tg[e] = tile[1][r + e][c]; tu[e] = tile[2][r + e][c]; }
        *(u32x4*)(actT + (int64_t)(c0 + c) * M + r0 + r) = *(const u32x4*)ta;
        *(u32x4*)(dguT + (int64_t)(c0 + c) * M + r0 + r) = *(const u32x4*)tg;
        *(u32x4*)(dguT + (int64_t)(I + c0 + c) * M + r0 + r) = *(const u32x4*)tu;
    }
}
__global__ __launch_bounds__(NT) void gelu_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t n, int kind) {
    for (int64_t i = (blockIdx.x * (int64_t)NT + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * NT * 8) {
        float f[8];
        unpack8(*(const u32x4*)(x + i), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = kind == MM355_GELU_ERF ? gelu_erf_f(f[e]) : gelu_tanh_f(f[e]);
        *(u32x4*)(y + i) = pack8(f);
    }
}
__global__ __launch_bounds__(NT) void gelu_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, int64_t n, int kind) {
    for (int64_t i = (blockIdx.x * (int64_t)NT + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * NT * 8) {
        float f[8], g[8];
        unpack8(*(const u32x4*)(x + i), f);
        unpack8(*(const u32x4*)(dy + i), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= kind == MM355_GELU_ERF ? gelu_erf_grad(f[e]) : gelu_tanh_grad(f[e]);
        *(u32x4*)(dx + i) = pack8(g);
    }
}

// ------------------------------------------------------------------------------------------------ scaling / casts
__global__ __launch_bounds__(NT) void scale_kernel(uint16_t* __restrict__ x, int64_t n, const float* __restrict__ s_dev, float s_host) {
    const float s = (s_dev ? *s_dev : 1.0f) * s_host;
    const int64_t nv = n >> 3;
    for (int64_t v = blockIdx.x * (int64_t)NT + threadIdx.x; v < nv; v += (int64_t)gridDim.x * NT) {
        float f[8];
        unpack8(*(const u32x4*)(x + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= s;
        *(u32x4*)(x + v * 8) = pack8(f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { const int64_t i = (nv << 3) + threadIdx.x; x[i] = f2bf(bf2f(x[i]) * s); }
}
template <typename TX>
__global__ __launch_bounds__(NT) void axpy_kernel(uint16_t* __restrict__ y, const TX* __restrict__ x, int64_t n, const float* __restrict__ s_dev,
                                                  float s_host, int accumulate) {
    const float s = (s_dev ? *s_dev : 1.0f) * s_host;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        float xv;
        if constexpr (sizeof(TX) == 2) xv = bf2f(x[i]); else xv = x[i];
        float r = s * xv;
        if (accumulate) r += bf2f(y[i]);
        y[i] = f2bf(r);
    }
}
__global__ __launch_bounds__(NT) void cast2d_kernel(const float* __restrict__ in, int64_t ld_in, uint16_t* __restrict__ out, int64_t ld_out, int64_t rows, int cols) {
    const int cv = cols >> 3;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t r = i / cv; const int c = (int)(i % cv) * 8;
        const f32x4 a = *(const f32x4*)(in + r * ld_in + c), b = *(const f32x4*)(in + r * ld_in + c + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        *(u32x4*)(out + r * ld_out + c) = pack8(f);
    }
}

// ------------------------------------------------------------------------------------------------ AdamW shard / grad norm
__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p32, float* __restrict__ m, float* __restrict__ v, const uint16_t* __restrict__ g,
                                                   uint16_t* __restrict__ pout, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2, const float* __restrict__ gs_dev) {
    const float gs = gs_dev ? *gs_dev : 1.0f;
    const float inv_sqrt_bc2 = rsqrtf(bc2);
    const float step = lr / bc1;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gr = bf2f(g[i]) * gs;
        float p = p32[i] * (1.0f - lr * wd);
        const float mi = m[i] * b1 + (1.0f - b1) * gr;
        const float vi = v[i] * b2 + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p -= step * (mi / denom);
        m[i] = mi; v[i] = vi; p32[i] = p;
        pout[i] = f2bf(p);
    }
}
// Grad-norm partials, deterministic: block b writes its sum to partials[b] (a fixed grid-stride walk, a fixed tree inside the block),
// sumsq_finish_kernel adds the partials in index order.  No atomics: the clip coefficient -- and through it every AdamW update -- is
// bit-identical from run to run.
__global__ __launch_bounds__(NT) void sumsq_kernel(const uint16_t* __restrict__ x, int64_t n, float* __restrict__ partials) {
    __shared__ float red[NT / 64];
    float s = 0.f;
    const int64_t nv = n >> 3;
    for (int64_t v = blockIdx.x * (int64_t)NT + threadIdx.x; v < nv; v += (int64_t)gridDim.x * NT) {
        float f[8];
        unpack8(*(const u32x4*)(x + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e] * f[e];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { const float f = bf2f(x[(nv << 3) + threadIdx.x]); s += f * f; }
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(NT) void sumsq_finish_kernel(const float* __restrict__ partials, int count, float* __restrict__ out) {
    __shared__ float red[NT / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < count; i += NT) s += partials[i];
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) out[0] += s;
}
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float pre, float* __restrict__ coef) {
    const float nrm = sqrtf(*sumsq);
    float c = max_norm > 0.f ? max_norm / (nrm + 1e-6f) : 1.0f;
    if (c > 1.0f) c = 1.0f;
    *coef = c * pre;
}

// ------------------------------------------------------------------------------------------------ im2col (SigLIP patch embedding)
template <typename TI>
__global__ __launch_bounds__(NT) void im2col_kernel(const TI* __restrict__ img, int N, int H, int W, int p, uint16_t* __restrict__ out, int Kp) {
    const int gh = H / p, gw = W / p, kreal = 3 * p * p;
    const int64_t total = (int64_t)N * gh * gw * Kp;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int k = (int)(i % Kp);
        const int64_t pr = i / Kp;
        uint16_t val = 0;
        if (k < kreal) {
            const int c = k / (p * p), dy = (k / p) % p, dx = k % p;
            const int px = (int)(pr % gw), py = (int)((pr / gw) % gh); const int64_t n = pr / ((int64_t)gw * gh);
            const TI x = img[((n * 3 + c) * H + (int64_t)py * p + dy) * W + (int64_t)px * p + dx];
            if constexpr (sizeof(TI) == 2) val = x; else val = f2bf(x);
        }
        out[i] = val;
    }
}

// ------------------------------------------------------------------------------------------------ splice gather / scatter
// one wave per output row
__global__ __launch_bounds__(NT) void splice_gather_kernel(const uint16_t* __restrict__ embed, const uint16_t* __restrict__ proj, const int32_t* __restrict__ src,
                                                           uint16_t* __restrict__ out, int64_t rows, int h) {
    const int64_t row = blockIdx.x * (int64_t)(NT / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, nv = h >> 3;
    const int s = src[row];
    const uint16_t* sp = s >= 0 ? embed + (int64_t)s * h : (s == -1 ? nullptr : proj + (int64_t)(-2 - s) * h);
    for (int v = lane; v < nv; v += 64)
        *(u32x4*)(out + row * h + v * 8) = sp ? *(const u32x4*)(sp + v * 8) : u32x4{0u, 0u, 0u, 0u};
}
// out[r] = idx[r] >= 0 ? in[idx[r]] : 0
__global__ __launch_bounds__(NT) void rows_gather_kernel(const uint16_t* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, uint16_t* __restrict__ out,
                                                         int64_t ld_out, int64_t R, int h) {
    const int64_t row = blockIdx.x * (int64_t)(NT / 64) + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63, nv = h >> 3;
    const int s = idx[row];
    for (int v = lane; v < nv; v += 64)
        *(u32x4*)(out + row * ld_out + v * 8) = s >= 0 ? *(const u32x4*)(in + (int64_t)s * ld_in + v * 8) : u32x4{0u, 0u, 0u, 0u};
}
__global__ __launch_bounds__(NT) void rows_scatter_add_kernel(const uint16_t* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ idx,
                                                              uint16_t* __restrict__ dst, int64_t ld_dst, int64_t R, int h) {
    const int64_t row = blockIdx.x * (int64_t)(NT / 64) + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63, nv = h >> 3;
    const int d = idx[row];
    if (d < 0) return;
    for (int v = lane; v < nv; v += 64) {
        float a[8], b[8];
        unpack8(*(const u32x4*)(src + row * ld_src + v * 8), a);
        unpack8(*(const u32x4*)(dst + (int64_t)d * ld_dst + v * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] += a[e];
        *(u32x4*)(dst + (int64_t)d * ld_dst + v * 8) = pack8(b);
    }
}
// embedding gradient, one wave per unique token id (segment)
__global__ __launch_bounds__(NT) void embed_grad_kernel(const uint16_t* __restrict__ dout, const int32_t* __restrict__ tok, const int32_t* __restrict__ seg,
                                                        const int32_t* __restrict__ pos, int64_t n_seg, uint16_t* __restrict__ dembed, int h, int accumulate) {
    const int64_t s = blockIdx.x * (int64_t)(NT / 64) + (threadIdx.x >> 6);
    if (s >= n_seg) return;
    const int lane = threadIdx.x & 63, nv = h >> 3;
    const int p0 = seg[s], p1 = seg[s + 1];
    uint16_t* drow = dembed + (int64_t)tok[s] * h;
    for (int v = lane; v < nv; v += 64) {
        float acc[8];
        if (accumulate) unpack8(*(const u32x4*)(drow + v * 8), acc);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        }
        for (int p = p0; p < p1; ++p) {
            float a[8];
            unpack8(*(const u32x4*)(dout + (int64_t)pos[p] * h + v * 8), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += a[e];
        }
        *(u32x4*)(drow + v * 8) = pack8(acc);
    }
}

}  // namespace

#define LAUNCH(kern, grid, ...) do { hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 0, (hipStream_t)stream, __VA_ARGS__); return mm_launch_status(); } while (0)

extern "C" int mm355_version(void) { return MM355_VERSION; }
extern "C" const char* mm355_strerror(int code) {
    switch (code) {
        case MM355_OK: return "ok";
        case MM355_EINVAL: return "invalid argument (pointer, dimension or alignment)";
        case MM355_EUNSUPPORTED: return "unsupported configuration";
        case MM355_ELAUNCH: return "HIP launch failure";
        default: return "unknown mm355 error";
    }
}

extern "C" int mm355_rope_table(mm355_bf16* cos_out, mm355_bf16* sin_out, int64_t L, int64_t d, float theta, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!cos_out || !sin_out || L <= 0 || d <= 0 || (d & 1)) return MM355_EINVAL;
    LAUNCH(rope_table_kernel, grid_for(L * d / 2), cos_out, sin_out, (int)L, (int)d, theta);
}
extern "C" int mm355_rope_table_freq(mm355_bf16* cos_out, mm355_bf16* sin_out, int64_t L, int64_t d, const float* inv_freq,
                                     float attention_scaling, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!cos_out || !sin_out || !inv_freq || L <= 0 || d <= 0 || (d & 1)) return MM355_EINVAL;
    LAUNCH(rope_table_freq_kernel, grid_for(L * d / 2), cos_out, sin_out, (int)L, (int)d, inv_freq, attention_scaling);
}
extern "C" int mm355_rope_qk(mm355_bf16* qkv, int64_t ld, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, const mm355_bf16* cos_t,
                             const mm355_bf16* sin_t, int inverse, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!qkv || !cos_t || !sin_t || B <= 0 || L <= 0 || Hq <= 0 || Hkv < 0 || d <= 0 || (d & 15) || (ld & 7)) return MM355_EINVAL;
    const int64_t H = Hq + Hkv;                              // q heads then k heads are contiguous column blocks
    LAUNCH(rope_qk_kernel, grid_for(B * L * H * (d / 16)), qkv, ld, (int)B, (int)L, (int)H, (int)d, cos_t, sin_t, inverse, (const int32_t*)nullptr);
}
extern "C" int mm355_rope_qk_pos(mm355_bf16* qkv, int64_t ld, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, const mm355_bf16* cos_t,
                                 const mm355_bf16* sin_t, const int32_t* pos_offset, int inverse, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!qkv || !cos_t || !sin_t || B <= 0 || L <= 0 || Hq <= 0 || Hkv < 0 || d <= 0 || (d & 15) || (ld & 7)) return MM355_EINVAL;
    const int64_t H = Hq + Hkv;
    LAUNCH(rope_qk_kernel, grid_for(B * L * H * (d / 16)), qkv, ld, (int)B, (int)L, (int)H, (int)d, cos_t, sin_t, inverse, pos_offset);
}
extern "C" int mm355_swiglu_fwd(const mm355_bf16* gu, mm355_bf16* act, int64_t M, int64_t I, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!gu || !act || M <= 0 || I <= 0 || (I & 7)) return MM355_EINVAL;
    LAUNCH(swiglu_fwd_kernel, grid_for(M * (I / 8)), gu, act, M, (int)I);
}
extern "C" int mm355_swiglu_bwd(const mm355_bf16* gu, const mm355_bf16* dact, mm355_bf16* dgu, mm355_bf16* act, int64_t M, int64_t I, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!gu || !dact || !dgu || M <= 0 || I <= 0 || (I & 7)) return MM355_EINVAL;
    LAUNCH(swiglu_bwd_kernel, grid_for(M * (I / 8)), gu, dact, dgu, act, M, (int)I);
}
extern "C" int mm355_swiglu_bwd_t(const mm355_bf16* gu, const mm355_bf16* dact, mm355_bf16* dgu, mm355_bf16* actT, mm355_bf16* dguT,
                                  int64_t M, int64_t I, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!gu || !dact || !dgu || !actT || !dguT || M <= 0 || I <= 0) return MM355_EINVAL;
    if ((M & 63) || (I & 63)) return MM355_EUNSUPPORTED;    // whole 64 x 64 tiles: else mm355_swiglu_bwd + mm355_transpose_bf16
    if (M / 64 > 65535) return MM355_EINVAL;
    hipLaunchKernelGGL(swiglu_bwd_t_kernel, dim3((unsigned)(I / 64), (unsigned)(M / 64)), dim3(256), 0, (hipStream_t)stream, gu, dact, dgu,
                       actT, dguT, M, (int)I);
    return mm_launch_status();
}
extern "C" int mm355_gelu_fwd(const mm355_bf16* x, mm355_bf16* y, int64_t n, int kind, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !y || n <= 0 || (n & 7)) return MM355_EINVAL;
    LAUNCH(gelu_fwd_kernel, grid_for(n / 8), x, y, n, kind);
}
extern "C" int mm355_gelu_bwd(const mm355_bf16* x, const mm355_bf16* dy, mm355_bf16* dx, int64_t n, int kind, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !dy || !dx || n <= 0 || (n & 7)) return MM355_EINVAL;
    LAUNCH(gelu_bwd_kernel, grid_for(n / 8), x, dy, dx, n, kind);
}
extern "C" int mm355_scale_bf16(mm355_bf16* x, int64_t n, const float* s_dev, float s_host, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || n <= 0 || !mm_aligned16(x)) return MM355_EINVAL;
    LAUNCH(scale_kernel, grid_for(n / 8 + 1), x, n, s_dev, s_host);
}
extern "C" int mm355_axpy_bf16(mm355_bf16* y, const mm355_bf16* x, int64_t n, const float* s_dev, float s_host, int accumulate, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !y || n <= 0) return MM355_EINVAL;
    LAUNCH(axpy_kernel<uint16_t>, grid_for(n), y, x, n, s_dev, s_host, accumulate);
}
extern "C" int mm355_axpy_f32_to_bf16(mm355_bf16* y, const float* x, int64_t n, float s_host, int accumulate, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !y || n <= 0) return MM355_EINVAL;
    LAUNCH(axpy_kernel<float>, grid_for(n), y, x, n, (const float*)nullptr, s_host, accumulate);
}
extern "C" int mm355_axpy_f32_to_bf16_dev(mm355_bf16* y, const float* x, int64_t n, const float* s_dev, float s_host, int accumulate, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !y || n <= 0) return MM355_EINVAL;
    LAUNCH(axpy_kernel<float>, grid_for(n), y, x, n, s_dev, s_host, accumulate);
}
extern "C" int mm355_cast_f32_bf16_2d(const float* in, int64_t ld_in, mm355_bf16* out, int64_t ld_out, int64_t rows, int64_t cols, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!in || !out || rows <= 0 || cols <= 0 || (cols & 7) || (ld_in & 3) || (ld_out & 7)) return MM355_EINVAL;
    LAUNCH(cast2d_kernel, grid_for(rows * (cols / 8)), in, ld_in, out, ld_out, rows, (int)cols);
}
extern "C" int mm355_adamw_shard(float* p32, float* m, float* v, const mm355_bf16* g, mm355_bf16* p_out, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, const float* grad_scale_dev,
                                 void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!p32 || !m || !v || !g || !p_out || n <= 0 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return MM355_EINVAL;
    LAUNCH(adamw_kernel, grid_for(n), p32, m, v, g, p_out, n, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale_dev);
}
extern "C" int mm355_sumsq_bf16(const mm355_bf16* x, int64_t n, float* out, float* partials, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!x || !out || !partials || n <= 0 || !mm_aligned16(x)) return MM355_EINVAL;
    const int blocks = (int)std::min<int64_t>((n / 8 + NT) / NT, MM355_SUMSQ_PARTIALS);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, n, partials);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, (const float*)partials, blocks, out);
    return mm_launch_status();
}
extern "C" int mm355_clip_coef(const float* sumsq, float max_norm, float pre_scale, float* coef, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!sumsq || !coef) return MM355_EINVAL;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, pre_scale, coef);
    return mm_launch_status();
}
extern "C" int mm355_im2col_patch(const void* images, int images_are_f32, int64_t N, int64_t H, int64_t W, int64_t p, mm355_bf16* out, int64_t Kp,
                                  void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!images || !out || N <= 0 || H < p || W < p || p <= 0 || Kp < 3 * p * p) return MM355_EINVAL;
    const int64_t total = N * (H / p) * (W / p) * Kp;
    if (images_are_f32) LAUNCH(im2col_kernel<float>, grid_for(total), (const float*)images, (int)N, (int)H, (int)W, (int)p, out, (int)Kp);
    LAUNCH(im2col_kernel<uint16_t>, grid_for(total), (const uint16_t*)images, (int)N, (int)H, (int)W, (int)p, out, (int)Kp);
}
extern "C" int mm355_splice_gather(const mm355_bf16* embed, const mm355_bf16* proj, const int32_t* src, mm355_bf16* out, int64_t rows, int64_t h,
                                   void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!embed || !src || !out || rows <= 0 || h <= 0 || (h & 7)) return MM355_EINVAL;
    LAUNCH(splice_gather_kernel, (unsigned)((rows + 3) / 4), embed, proj, src, out, rows, (int)h);
}
extern "C" int mm355_rows_gather(const mm355_bf16* in, int64_t ld_in, const int32_t* idx, mm355_bf16* out, int64_t ld_out, int64_t R, int64_t h,
                                 void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!in || !idx || !out || R <= 0 || h <= 0 || (h & 7) || (ld_in & 7) || (ld_out & 7)) return MM355_EINVAL;
    LAUNCH(rows_gather_kernel, (unsigned)((R + 3) / 4), in, ld_in, idx, out, ld_out, R, (int)h);
}
extern "C" int mm355_rows_scatter_add(const mm355_bf16* src, int64_t ld_src, const int32_t* idx, mm355_bf16* dst, int64_t ld_dst, int64_t R,
                                      int64_t h, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!src || !idx || !dst || R <= 0 || h <= 0 || (h & 7) || (ld_src & 7) || (ld_dst & 7)) return MM355_EINVAL;
    LAUNCH(rows_scatter_add_kernel, (unsigned)((R + 3) / 4), src, ld_src, idx, dst, ld_dst, R, (int)h);
}
extern "C" int mm355_embed_grad(const mm355_bf16* dout, const int32_t* tok, const int32_t* seg_start, const int32_t* pos, int64_t n_seg,
                                mm355_bf16* dembed, int64_t h, int accumulate, void* stream) {
    (void)hipGetLastError();   // drop any stale, unrelated runtime status before we launch
    if (!dout || !tok || !seg_start || !pos || !dembed || n_seg <= 0 || h <= 0 || (h & 7)) return MM355_EINVAL;
    LAUNCH(embed_grad_kernel, (unsigned)((n_seg + 3) / 4), dout, tok, seg_start, pos, n_seg, dembed, (int)h, accumulate);
}
