// Shared argument block of the second-generation attention kernels (attn2.hip), launched from attn.hip.
#pragma once
#include "mm355_common.h"

namespace attn2 {
struct Args {
    const uint16_t* q; const uint16_t* k; const uint16_t* v; const uint16_t* d_o;
    int64_t ld_q, ld_k, ld_o;
    uint16_t* o; float* lse; const float* lse_in; const float* delta; uint16_t* dqb; int64_t ld_dq; const int32_t* seqlens;
    uint16_t* dk; uint16_t* dv; float* dkp; float* dvp; int64_t ld_dkv;   // dK/dV kernel outputs (bf16 direct, or fp32 per-query-head partials)
    int B, L, Hq, Hkv, d;
    float scale; int causal;
    // backward only, d == 128 kernels: when set, dq / dk leave the kernels already rotated back through RoPE (the inverse rotation of
    // mm355_rope_qk on the bf16-rounded gradients): cos / sin tables [positions][128] bf16, optional per-sample position offsets
    const uint16_t* rope_cos; const uint16_t* rope_sin; const int32_t* rope_pos;
    // diagnostics (mm355_attn_fwd_debug; nullptr in the product path): attn4::fwd_kernel writes dbg[((b * Hq + hq) * ceil(L / 256) + query block) * 4 + wave] = how many times
    // that wave took the deferred-rescale branch (running maximum moved because some row grew by more than 2^THR)
    int32_t* dbg;
};

#ifdef __HIPCC__
constexpr int NT = 256;

MM_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int DP> struct Geo {
    static constexpr int DS = DP == 64 ? 64 : 128;          // LDS row length (elements) of a [rows][d] tile
    static constexpr int KS = DP / 32;                      // k-steps over d
    static constexpr int NF = DP / 16;                      // 16-wide fragments over d
    static constexpr int T64 = 64 * DS * 2;                 // bytes of a [64][DS] tile
    static constexpr int V64 = 64 * (DP / 8);               // 16-B vectors of a [64][DP] tile
};

MM_DEV bf16x8 pack_acc(const f32x4& a, const f32x4& b) {
    u32x4 w;
    w.x = pack2bf(a[0], a[1]); w.y = pack2bf(a[2], a[3]); w.z = pack2bf(b[0], b[1]); w.w = pack2bf(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, w);
}

template <int DS> MM_DEV int swzN(int row) { return DS == 128 ? 2 * (row & 7) : 2 * ((row >> 1) & 3); }
template <int DS> MM_DEV int offN(int row, int chunk) { return row * (DS * 2) + ((chunk ^ swzN<DS>(row)) << 4); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds4_t;

// X[row0 + fq*4 + e][c16*16 + fr] (e < 4) and X[row0 + 16 + fq*4 + e - 4][c16*16 + fr] (e >= 4) from a natural tile
template <int DS>
MM_DEV bf16x8 read_nat_perm(const unsigned char* s, int row0, int c16, int fr, int fq) {
#ifdef MM355_ABLATE_TR        // TIMING-ONLY ablation build (wrong results): what would a ds_read_b128 in place of the two tr_b64 gathers buy?
    return *(const bf16x8*)(s + offN<DS>(row0 + fr, c16 * 2 + (fq & 1)) + (fq >> 1) * 8 * DS * 2 * 0);
#endif
    const int j = fr >> 2, q4 = fr & 3;
    const int chunk = c16 * 2 + (q4 >> 1), sub = (q4 & 1) * 8;
    const int r_lo = row0 + fq * 4 + j, r_hi = r_lo + 16;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(s + offN<DS>(r_lo, chunk) + sub));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(s + offN<DS>(r_hi, chunk) + sub));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// The same gather with the lane-dependent part of the address hoisted: nat_perm_off() is loop-invariant per 16-column group c16
// (one VGPR each); a tile row block row0 (a multiple of 8, so the swizzle phase is unchanged) and the +16-row half are
// immediates of the two reads.
template <int DS> MM_DEV int nat_perm_off(int c16, int fr, int fq) {
    const int j = fr >> 2, q4 = fr & 3;
    return offN<DS>(fq * 4 + j, c16 * 2 + (q4 >> 1)) + (q4 & 1) * 8;
}
template <int DS> MM_DEV bf16x8 read_nat_perm_at(const unsigned char* s, int row0, int off) {
    const unsigned char* p = s + row0 * (DS * 2) + off;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p + 16 * DS * 2));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// all-reduce over the four lanes {l, l^16, l^32, l^48} that share one query column, on the VALU (v_permlane*_swap), no LDS
MM_DEV void swap16(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __uint_as_float(r0); b = __uint_as_float(r1);
}
MM_DEV void swap32(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __uint_as_float(r0); b = __uint_as_float(r1);
}
// single-instruction maxima: fmaxf() on MFMA outputs makes hipcc emit a canonicalising v_max x, x in front of every operand
// (IEEE sNaN quieting) -- three instructions where one does; scores are finite or -inf here
MM_DEV float max2_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
MM_DEV float max3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
MM_DEV float quad_max(float x) {
    float a = x, b = x;
    asm volatile("" : "+v"(b));
    swap16(a, b);
    a = max2_raw(a, b); b = a;
    asm volatile("" : "+v"(b));
    swap32(a, b);
    return max2_raw(a, b);
}
MM_DEV float quad_sum(float x) {
    float a = x, b = x;
    asm volatile("" : "+v"(b));
    swap16(a, b);
    a = a + b; b = a;
    asm volatile("" : "+v"(b));
    swap32(a, b);
    return a + b;
}
#endif
}  // namespace attn2

int mm355_attn2_fwd_launch(const attn2::Args& a, int dp, hipStream_t s);
int mm355_attn2_dq_launch(const attn2::Args& a, int dp, hipStream_t s);
int mm355_attn2_dkdv_launch(const attn2::Args& a, int dp, hipStream_t s);
// d == 128 fast paths (attn3.hip): LDS-DMA staged, double-buffered
int mm355_attn3_fwd_launch(const attn2::Args& a, hipStream_t s);
int mm355_attn3_dq_launch(const attn2::Args& a, hipStream_t s);
int mm355_attn3_dkdv_launch(const attn2::Args& a, hipStream_t s);
// d == 128 forward, one wave per SIMD, hand-placed stream (attn4.hip); variant 1 = the serialised debugging stream
int mm355_attn4_fwd_launch(const attn2::Args& a, int variant, hipStream_t s);
// d == 128 backward (dK / dV then dQ), same construction (attn4_bwd.hip); workspace: 2 * B * Hq * L floats
int mm355_attn4_bwd_launch(const attn2::Args& a, float* workspace, int variant, hipStream_t s);
