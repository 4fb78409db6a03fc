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
};
}  // namespace attn2

int mm355_attn2_fwd_launch(const attn2::Args& a, int dp, hipStream_t s);
int mm355_attn2_dq_launch(const attn2::Args& a, int dp, hipStream_t s);
int mm355_attn2_dkdv_launch(const attn2::Args& a, int dp, hipStream_t s);
