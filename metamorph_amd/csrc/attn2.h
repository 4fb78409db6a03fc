// Shared argument block of the second-generation attention kernels (attn2.hip), launched from attn.hip.
#pragma once
#include "mm355_common.h"

namespace attn2 {
struct Args {
    const uint16_t* q; const uint16_t* k; const uint16_t* v; const uint16_t* vt; const uint16_t* kt; const uint16_t* d_o;
    int64_t ld_q, ld_k, ld_o;
    uint16_t* o; float* lse; const float* lse_in; const float* delta; float* dq; const int32_t* seqlens;
    int B, L, Lp, Hq, Hkv, d;
    float scale; int causal;
};
}  // namespace attn2

int mm355_attn2_fwd_launch(const attn2::Args& a, int dp, hipStream_t s);
int mm355_attn2_dq_launch(const attn2::Args& a, int dp, hipStream_t s);
