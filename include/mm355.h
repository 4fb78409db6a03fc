/*
 * mm355.h -- C ABI of libmm355.so: the MI355X (gfx950 / CDNA4) kernels behind the MetaMorph hot path.
 *
 * The reference (facebookresearch/metamorph) has NO FFI layer: its boundary is the Python class
 * contract of metamorph.model and all device arithmetic is reached through torch /
 * transformers.  This header is therefore the ABI a maintainer would bind FROM the reference's Python
 * (ctypes); every entry point cites the reference call site whose arithmetic it
 * replaces.  Paths are relative to the reference checkout; "HF" = transformers (pinned 4.45.0 in
 * pyproject.toml:16, not vendored).
 *
 * Conventions
 *   - raw DEVICE pointers, caller allocates everything (no hipMalloc / hipFree / sync inside);
 *   - row-major, leading dimensions in ELEMENTS; bf16 is the storage type unless a name says f32;
 *   - 16-byte aligned base pointers and leading dimensions that are multiples of 8 elements;
 *   - asynchronous on `stream` (a hipStream_t passed as void*);
 *   - returns 0 or a negative MM355_E* code; never throws, never exits;
 *   - re-entrant; the only process-wide state is one-time kernel attribute setup (LDS size caps) -- no environment variable is read,
 *     nothing depends on the call sequence.
 */
#ifndef MM355_H
#define MM355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM355_VERSION 100            /* 0.1.0 */

#define MM355_OK            0
#define MM355_EINVAL       -1        /* bad pointer / dimension / alignment                */
#define MM355_EUNSUPPORTED -2        /* legal request this build has no kernel for         */
#define MM355_ELAUNCH      -3        /* hipLaunch reported an error (hipGetLastError)      */

typedef uint16_t mm355_bf16;         /* raw bfloat16 bits                                  */

int mm355_version(void);
const char* mm355_strerror(int code);

/* ------------------------------------------------------------------------------------------------
 * GEMM family.   C[M,N] = epilogue( A[M,K] . B[N,K]^T )      (y = x W^T, the nn.Linear form)
 * Replaces every nn.Linear reached on the path: HF LlamaAttention/LlamaMLP projections
 * (metamorph_llama.py:349-359), lm_head (:398), mm_projector (multimodal_projector/builder.py:52-59),
 * vision_head (metamorph_llama.py:252-256), SigLIP q/k/v/out/fc1/fc2 (siglip_encoder.py:141) and the
 * patch-embedding Conv2d lowered to a GEMM.  Backward GEMMs (dX = dY W, dW = dY^T X) use the same
 * entry point on transposed operands (mm355_transpose_bf16).
 *   epilogue: v = acc; if BIAS v += bias[n]; if GELU_* v = gelu(v); if RESIDUAL v += R[m % res_mod][n];
 *             if ACCUMULATE v += C_old[m][n];  store as bf16 (or f32 with OUT_F32).
 * Requirements: K % 8 == 0, lda/ldb % 8 == 0 (ldc/ldr % 8 == 0 for bf16 vector stores, else scalar tail).
 * variant: 0 = auto (ping-pong 256x256 kernel, variant 11, once >= 200 tiles and K % 64 == 0; 128x128 LDS-DMA otherwise);
 *          1..mm355_gemm_num_variants() select a specific tile configuration / schedule (bench / tests); 13 = the one-wave-per-SIMD
 *          kernel (csrc/gemm_st.hip: persistent 4-wave workgroups, 128 x 128 wave tiles, hand-placed stream; K % 128 == 0, K >= 256,
 *          MM355_EUNSUPPORTED otherwise), 14 = the same stream serialised -- both bit-identical to 11.
 * ------------------------------------------------------------------------------------------------ */
#define MM355_GEMM_BIAS        1u
#define MM355_GEMM_GELU_ERF    2u    /* nn.GELU() default (projector, vision_head)         */
#define MM355_GEMM_GELU_TANH   4u    /* SigLIP "gelu_pytorch_tanh"                          */
#define MM355_GEMM_RESIDUAL    8u
#define MM355_GEMM_ACCUMULATE 16u
#define MM355_GEMM_OUT_F32    32u

int mm355_gemm_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb,
                    void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    const mm355_bf16* bias, const mm355_bf16* residual, int64_t ldr, int64_t res_row_mod,
                    uint32_t flags, int variant, void* stream);
int mm355_gemm_num_variants(void);

/* Split-K form for PROMPT-PASS shapes (reference: the same nn.Linear calls, reached with a few hundred rows when `generate` runs the prompt,
 * metamorph_llama.py:665-717): C[M][N] = bf16(A . B^T (+ residual)) with K cut into slices that run as separate workgroups of ONE launch (64 x
 * 128 tiles x slices; fp32 partials in `workspace`, summed in slice order by a second launch).  At M <= ~1000 against N = 4096 .. 6144 the plain
 * kernels are a latency chain of one LDS-DMA round trip per K tile (down_proj, K = 14336: 184 us at any such M); the slices run side by side.
 * Not bit-identical to mm355_gemm_bf16 (another fp32 summation order).  workspace: mm355_gemm_splitk_ws_floats(M, N, K) floats (0: the shape
 * is not split -- the call then forwards to mm355_gemm_bf16 and needs none).  K % 64 == 0, N % 8 == 0, leading dimensions % 8 == 0. */
int64_t mm355_gemm_splitk_ws_floats(int64_t M, int64_t N, int64_t K);
int mm355_gemm_splitk_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb, mm355_bf16* C, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, const mm355_bf16* residual, int64_t ldr, float* workspace,
                           int64_t workspace_floats, void* stream);
/* The split projection with the launch that FOLLOWS it on the decode / prompt path folded into the reduce launch (a launch costs 3 - 5 us whatever
 * it does; the decode step of more than 16 sequences had thirteen per layer, nine with these).  Each form rounds the slice sums to bf16 where
 * mm355_gemm_splitk_bf16 stores them and continues with the arithmetic of the kernel it replaces: the bits of the launch sequence.  A shape that
 * is not split runs that sequence inside the library.  Reference: HF LlamaDecoderLayer at decode shape, metamorph_llama.py:665-717.
 *   _norm:        C[M][N] = bf16(A . B^T + residual) (rows N apart), Y[M][N] = RMSNorm(C; norm_w, eps)   == gemm_splitk -> rmsnorm_fwd
 *                 (o projection -> post-attention norm; down projection -> the next layer's input norm).  workspace: mm355_gemm_splitk_ws_floats.
 *   _swiglu:      act[M][I] = SiLU(g) * u, [g | u] = X . Wgu[2 I][K]^T                                    == gemm_splitk -> swiglu_fwd
 *                 workspace: mm355_gemm_splitk_swiglu_ws_floats(M, I, K) floats (never 0: the unsplit sequence parks its bf16 g | u rows there).
 *   _rope_append: one new q|k|v row per sequence: q rotated at positions[m] (device) -> qkv[m][0 .. Hq d), rotated k and v -> cache row
 *                 positions[m]; the k | v columns of qkv are not written                                   == gemm_splitk -> rope_kv_append
 *                 workspace: mm355_gemm_splitk_ws_floats(M, (Hq + 2 Hkv) d, K).  M <= 65535, d % 16 == 0. */
int mm355_gemm_splitk_norm_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* B, int64_t ldb, mm355_bf16* C, int64_t M, int64_t N,
                                int64_t K, const mm355_bf16* residual, int64_t ldr, const mm355_bf16* norm_w, float eps, mm355_bf16* Y,
                                float* workspace, int64_t workspace_floats, void* stream);
int64_t mm355_gemm_splitk_swiglu_ws_floats(int64_t M, int64_t I, int64_t K);
int mm355_gemm_splitk_swiglu_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wgu, int64_t ldw, mm355_bf16* act, int64_t ld_act,
                                  int64_t M, int64_t I, int64_t K, float* workspace, int64_t workspace_floats, void* stream);
int mm355_gemm_splitk_rope_append_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wqkv, int64_t ldw, mm355_bf16* qkv, int64_t ld_qkv,
                                       int64_t M, int64_t Hq, int64_t Hkv, int64_t d, int64_t K, const mm355_bf16* cos_t,
                                       const mm355_bf16* sin_t, const int32_t* positions, mm355_bf16* k_cache, mm355_bf16* v_cache,
                                       int64_t ld_kv, int64_t batch_stride_kv, float* workspace, int64_t workspace_floats, void* stream);

/* Fused gate|up projection + SwiGLU of the LLaMA MLP (reference: HF LlamaMLP `down_proj(act_fn(gate_proj(x)) * up_proj(x))`, reached at
 * metamorph_llama.py:349-359):  gu[M][2 I] = X[M][K] . Wgu[2 I][K]^T  (gate columns 0..I-1, up columns I..2I-1, exactly what
 * mm355_gemm_bf16 writes) AND act[M][I] = bf16(silu(gate)) * up from the bf16-rounded gu values (exactly what mm355_swiglu_fwd writes), in
 * ONE launch: a workgroup's 256 tile columns are 128 gate channels and the same 128 up channels, so the product is formed in the epilogue
 * registers and the separate read of gu disappears.  Requirements: I % 128 == 0, K % 128 == 0, leading dimensions % 8 == 0;
 * MM355_EUNSUPPORTED otherwise (callers fall back to mm355_gemm_bf16 + mm355_swiglu_fwd: same bits). */
int mm355_gemm_swiglu_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wgu, int64_t ldw, mm355_bf16* gu, int64_t ld_gu,
                           mm355_bf16* act, int64_t ld_act, int64_t M, int64_t I, int64_t K, void* stream);

/* Fused q|k|v projection + RoPE (reference: HF LlamaAttention q_proj / k_proj / v_proj + apply_rotary_pos_emb, reached at
 * metamorph_llama.py:349-359), head size 128:  qkv[M][N] = X[M][K] . Wqkv[N][K]^T with the rotate-half rotation of mm355_rope_qk applied
 * in the epilogue to the columns below n_rot (= (Hq + Hkv) * 128: the q and k blocks; the v block is stored unrotated) -- the bits of
 * mm355_gemm_bf16 followed by mm355_rope_qk[_pos], without the in-place pass over q and k.  Row r is position r % L (+ pos_offset[r / L]
 * when pos_offset != NULL) of the cos / sin tables ([positions][128] bf16 from mm355_rope_table).  Requirements: N % 256 == 0,
 * n_rot % 128 == 0, K % 128 == 0, leading dimensions % 8 == 0; MM355_EUNSUPPORTED otherwise. */
int mm355_gemm_rope_bf16(const mm355_bf16* X, int64_t ldx, const mm355_bf16* Wqkv, int64_t ldw, mm355_bf16* qkv, int64_t ld_qkv,
                         const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* pos_offset,
                         int64_t M, int64_t N, int64_t K, int64_t L, int64_t n_rot, void* stream);

/* Fused backward of the same MLP stage: d act[M][I] = dY[M][K] . WdT[I][K]^T (the down_proj input gradient; WdT = down_proj.weight
 * transposed, [I][K]) is formed in the accumulators, rounded to bf16, and fed straight into the SwiGLU backward with gate / up from
 * gu[M][2 I]:  dgu[M][2 I] (row-major: the operand of the gate|up input-gradient GEMM) AND the contraction-major copies actT[I][ldT],
 * dguT[2 I][ldT] (the operands of the down_proj / gate|up weight-gradient GEMMs) -- the outputs of mm355_gemm_bf16 + mm355_swiglu_bwd_t,
 * without d act ever reaching memory.  Requirements: I % 64 == 0, M % 8 == 0, K % 128 == 0, ldT >= M, leading dimensions % 8 == 0;
 * MM355_EUNSUPPORTED otherwise.  Columns [M, ldT) of actT / dguT are not written (callers zero their padding). */
int mm355_gemm_swiglu_bwd_bf16(const mm355_bf16* dY, int64_t ldy, const mm355_bf16* WdT, int64_t ldw, const mm355_bf16* gu, int64_t ld_gu,
                               mm355_bf16* dgu, int64_t ld_dgu, mm355_bf16* actT, mm355_bf16* dguT, int64_t ldT,
                               int64_t M, int64_t I, int64_t K, void* stream);

/* Two independent problems of the form above, C0 (+)= A0 . B0^T and C1 (+)= A1 . B1^T, in ONE launch of the 256x256 ping-pong
 * kernel.  A launch runs in waves of 256 workgroups (one tile per CU): the LLaMA-3-8B weight gradients of qkv (384 tiles) and
 * down_proj (896 tiles) cost 2 + 4 wave times launched separately and 5 as a pair.  The host side pairs them at the end of
 * DecoderLayerFn.backward (reference: the two nn.Linear weight gradients autograd computes in LlamaDecoderLayer's backward,
 * call site metamorph_llama.py:349-359).
 * flags per problem: MM355_GEMM_ACCUMULATE, MM355_GEMM_OUT_F32.  Requirements per problem: K % 128 == 0, lda/ldb % 8 == 0,
 * 256 rows x ld x 2 B below 2 GiB (else MM355_EUNSUPPORTED: launch them one by one with mm355_gemm_bf16). */
int mm355_gemm_pair_bf16(const mm355_bf16* A0, int64_t lda0, const mm355_bf16* B0, int64_t ldb0, void* C0, int64_t ldc0,
                         int64_t M0, int64_t N0, int64_t K0, uint32_t flags0,
                         const mm355_bf16* A1, int64_t lda1, const mm355_bf16* B1, int64_t ldb1, void* C1, int64_t ldc1,
                         int64_t M1, int64_t N1, int64_t K1, uint32_t flags1, void* stream);

/* Weight-gradient form on the operands AS THEY LIE IN MEMORY (no transposed copies):
 *   C[M,N] (+)= At[K,M]^T . Bt[K,N]      e.g. dW[out,in] = dY[tokens,out]^T . X[tokens,in]
 * MFMA fragments are gathered from contraction-major LDS tiles with ds_read_b64_tr_b16.
 * Requirements: K % 64 == 0 (else MM355_EUNSUPPORTED: use mm355_transpose_bf16 + mm355_gemm_bf16), M, N, lda, ldb % 8 == 0.
 * flags: MM355_GEMM_ACCUMULATE, MM355_GEMM_OUT_F32. */
int mm355_gemm_tn_bf16(const mm355_bf16* At, int64_t lda, const mm355_bf16* Bt, int64_t ldb, void* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, uint32_t flags, void* stream);

/* Input-gradient form on the weight AS IT LIES IN MEMORY:
 *   C[M,N] = A[M,K] . Bt[K,N] (+ residual)      e.g. dX[tokens,in] = dY[tokens,out] . W[out,in]
 * A fragments by ds_read_b128, Bt fragments by ds_read_b64_tr_b16 (ping-pong 256x256 kernel).
 * Requirements: K % 128 == 0, N, lda, ldb, ldc % 8 == 0, operands below 2 GiB (else MM355_EUNSUPPORTED: use
 * mm355_transpose_bf16 + mm355_gemm_bf16).  flags: MM355_GEMM_RESIDUAL, MM355_GEMM_ACCUMULATE, MM355_GEMM_OUT_F32. */
int mm355_gemm_nn_bf16(const mm355_bf16* A, int64_t lda, const mm355_bf16* Bt, int64_t ldb, void* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, const mm355_bf16* residual, int64_t ldr,
                       uint32_t flags, void* stream);

/* out[c][r] = in[r][c]   (rows x cols -> cols x rows), bf16.  Used for the backward GEMM operands.
 * ld_in % 8 == 0, in / out 16-byte aligned, ld_out >= rows (else MM355_EINVAL: output rows would overlap). */
int mm355_transpose_bf16(const mm355_bf16* in, int64_t ld_in, int64_t rows, int64_t cols,
                         mm355_bf16* out, int64_t ld_out, void* stream);

/* column sums: db[n] += sum_m dY[m][n]  (bias gradients of mm_projector / vision_head).  Deterministic: one workgroup owns 64 columns
 * and all M rows (four row phases, fixed-order reduction), no atomics. */
int mm355_colsum_bf16(const mm355_bf16* dY, int64_t ld, int64_t M, int64_t N, float* db_f32, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RMSNorm -- HF LlamaRMSNorm (fp32 normalise, cast to bf16, THEN multiply by weight); K7.
 * bwd: dx[m] = (dres ? dres[m] : 0) + d/dx ; dw_f32[h] += sum_m dy*xhat (caller zeroes dw_f32).
 * workspace (optional, mm355_rmsnorm_bwd_ws_floats(M, h) floats, 16-B aligned): the weight gradient is then summed
 * from per-workgroup partial rows in a fixed order (deterministic, no atomics); NULL = fp32 atomics on dw_f32.
 * ------------------------------------------------------------------------------------------------ */
int mm355_rmsnorm_fwd(const mm355_bf16* x, const mm355_bf16* w, mm355_bf16* y, int64_t M, int64_t h,
                      float eps, void* stream);
/* the same, also returning rstd[m] = rsqrt(mean(x[m]^2) + eps) (fp32, M values; NULL = not wanted) ... */
int mm355_rmsnorm_fwd_rstd(const mm355_bf16* x, const mm355_bf16* w, mm355_bf16* y, float* rstd_out, int64_t M, int64_t h,
                           float eps, void* stream);
/* ... from which the backward pass takes the TRANSPOSED normalised activations in one pass:
 *   out_t[c][m] = w[c] * bf16(x[m][c] * rstd[m]),   out_t is [h][ld_out], ld_out >= M (columns M..ld_out are not written)
 * = the contraction-major operand of the qkv / gate_up weight-gradient GEMMs (HF computes those from the saved norm output,
 * reference call site metamorph_llama.py:349-359); bit-identical to transposing mm355_rmsnorm_fwd's result. */
int mm355_rmsnorm_apply_t(const mm355_bf16* x, const mm355_bf16* w, const float* rstd, int64_t M, int64_t h,
                          mm355_bf16* out_t, int64_t ld_out, void* stream);
int mm355_rmsnorm_bwd(const mm355_bf16* dy, const mm355_bf16* x, const mm355_bf16* w,
                      const mm355_bf16* dres, mm355_bf16* dx, float* dw_f32, float* workspace,
                      int64_t M, int64_t h, float eps, void* stream);
int64_t mm355_rmsnorm_bwd_ws_floats(int64_t M, int64_t h);
/* the same with the weight gradient landing straight in the parameter's bf16 gradient buffer (training hot path: no fp32
 * scratch to zero, no separate accumulate pass): w_grad[h] = (accumulate ? w_grad : 0) + sum_m dy*xhat, fixed summation order;
 * workspace (mm355_rmsnorm_bwd_ws_floats floats) is required. */
int mm355_rmsnorm_bwd_wgrad(const mm355_bf16* dy, const mm355_bf16* x, const mm355_bf16* w, const mm355_bf16* dres,
                            mm355_bf16* dx, mm355_bf16* w_grad, int accumulate, float* workspace,
                            int64_t M, int64_t h, float eps, void* stream);

/* LayerNorm forward (SigLIP encoder, eps 1e-6); the tower is frozen in every shipped recipe. */
int mm355_layernorm_fwd(const mm355_bf16* x, const mm355_bf16* w, const mm355_bf16* b, mm355_bf16* y,
                        int64_t M, int64_t h, float eps, void* stream);
/* LayerNorm backward (trainable tower; HF SiglipEncoderLayer.layer_norm1/2 under freeze_vision=False,
 * siglip_encoder.py:138-141): dx = (dres ? dres : 0) + d/dx; dw_f32[h] += sum dy*xhat; db_f32[h] += sum dy, both summed in a
 * fixed order from per-workgroup partial rows in `workspace` (mm355_layernorm_bwd_ws_floats(M, h) floats, 16-B aligned). */
int64_t mm355_layernorm_bwd_ws_floats(int64_t M, int64_t h);
int mm355_layernorm_bwd(const mm355_bf16* dy, const mm355_bf16* x, const mm355_bf16* w, const mm355_bf16* dres,
                        mm355_bf16* dx, float* dw_f32, float* db_f32, float* workspace,
                        int64_t M, int64_t h, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RoPE -- HF apply_rotary_pos_emb, rotate-half convention, theta from config; K9.
 * Tables: cos/sin[L][d] bf16 (fp32 math, then cast -- LlamaRotaryEmbedding.forward).
 * mm355_rope_qk rotates the q and k column blocks of a fused qkv activation [B*L, ld] IN PLACE
 * (q heads at column 0, k heads at column Hq*d).  inverse != 0 applies the transposed rotation
 * (backward).  Position of row (b,l) is l.
 * ------------------------------------------------------------------------------------------------ */
int mm355_rope_table(mm355_bf16* cos_out, mm355_bf16* sin_out, int64_t L, int64_t d, float theta, void* stream);
/* Tables of a SCALED RoPE (config.rope_scaling / rope_parameters -- LLaMA-3.1's rope_type "llama3", "linear"; reference: MetaMorphConfig
 * inherits the field from LlamaConfig, metamorph_llama.py:129-133, and HF's LlamaRotaryEmbedding, reached at :349-359, rescales inv_freq
 * per wavelength band once at construction): inv_freq = float[d/2] on the DEVICE, computed by the caller exactly as
 * ROPE_INIT_FUNCTIONS[rope_type] does; table[l][j] = table[l][d/2 + j] = bf16(cos|sin(l * inv_freq[j]) * attention_scaling).  Every
 * consumer of the tables (mm355_rope_qk*, mm355_gemm_rope_bf16, the attention backward epilogues, the decode kernels) is unchanged. */
int mm355_rope_table_freq(mm355_bf16* cos_out, mm355_bf16* sin_out, int64_t L, int64_t d, const float* inv_freq, float attention_scaling,
                          void* stream);
int mm355_rope_qk(mm355_bf16* qkv, int64_t ld, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d,
                  const mm355_bf16* cos_t, const mm355_bf16* sin_t, int inverse, void* stream);
/* the same with a per-sample position offset (int32[B], device): position of row (b,l) is l + pos_offset[b]; the tables need
 * L + max(pos_offset) rows.  Left-padded batches (tokenizer_padding_side = "left", metamorph_arch.py:362-386) are run right-aligned
 * to row 0 with pos_offset[b] = number of padding rows, which reproduces HF's position ids (arange(L), padding included). */
int mm355_rope_qk_pos(mm355_bf16* qkv, int64_t ld, int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d,
                      const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* pos_offset, int inverse, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention -- torch SDPA as driven by HF LlamaModel (causal + key padding, GQA, fp32 softmax; K10)
 * and by HF SiglipAttention (non causal, d = 72; K2).
 * q/k/v are column blocks of row-major activations: element (b,l,head,dd) at ptr[(b*L + l)*ld + head*d + dd]
 * (k and v share ld_k).  No transposed copies are needed: the kernels gather contraction-over-rows operands with
 * ds_read_b64_tr_b16.  seqlens[b] = number of valid (non padding) keys of sample b (right padding); NULL = L.
 * o: [B*L][>= Hq*d] bf16; lse: [B][Hq][L] f32 (natural-log sum-exp of the scaled scores).
 * Rows l >= seqlens[b] produce o = 0, lse = 0 (and zero gradients).
 * Supported d: any d % 8 == 0 and d <= 128 (64 / 128 LLaMA, 72 SigLIP SO400M).
 * ------------------------------------------------------------------------------------------------ */
int mm355_attn_fwd(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k,
                   mm355_bf16* o, int64_t ld_o, float* lse, const int32_t* seqlens,
                   int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, float scale, int causal, void* stream);

/* mm355_attn_fwd with the kernel generation chosen by the caller -- for tests and tools/ (A/B timing, the serialised debugging stream);
 * the product calls mm355_attn_fwd.  variant: 0 = as mm355_attn_fwd; 2 = the generic-d kernels; 3 = the round-2 two-waves-per-SIMD d == 128 kernels
 * (-DMM355_LEGACY_VARIANTS builds only); 4 = the one-wave-per-SIMD
 * hand-placed stream (d == 128); 41 = the same stream serialised (every LDS read waited for at once, 32 wait states behind every MFMA:
 * bit-identical to 4 by construction).  MM355_EUNSUPPORTED when the variant does not cover the geometry.  Same reference call site as
 * mm355_attn_fwd (torch SDPA reached at metamorph_llama.py:349-359). */
int mm355_attn_fwd_variant(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k,
                           mm355_bf16* o, int64_t ld_o, float* lse, const int32_t* seqlens,
                           int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, float scale, int causal, int variant, void* stream);

/* Diagnostics of the d == 128 stream (tests only; the product calls mm355_attn_fwd): mm355_attn_fwd_variant with variant 4 / 41 that also
 * reports how often each wave took the DEFERRED-RESCALE branch -- the kernel keeps a stale running row maximum m and rescales O and the
 * row sums only when some row of the wave's 64 query rows grew by more than 2^6 over it.  rescale_counts: int32 [B][Hq][ceil(L / 256)][4],
 * zeroed by the caller; entry = number of key tiles (of 64) at which that wave moved its maxima.  The parity tests on adversarial score
 * distributions (tests/test_attn_hostile_gpu.py) assert these counts against a CPU model of the kernel's decision rule, so that "the
 * branch ran" is a measured fact and not an assumption.  MM355_EUNSUPPORTED unless d == 128.  Same reference call site as mm355_attn_fwd
 * (torch SDPA reached at metamorph_llama.py:349-359). */
int mm355_attn_fwd_debug(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k,
                         mm355_bf16* o, int64_t ld_o, float* lse, const int32_t* seqlens,
                         int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, float scale, int causal, int variant,
                         int32_t* rescale_counts, void* stream);

/* delta[b][h][l] = sum_dd dO*O  (softmax-backward row term). */
int mm355_attn_bwd_prep(const mm355_bf16* o, const mm355_bf16* d_o, int64_t ld_o, float* delta,
                        int64_t B, int64_t L, int64_t Hq, int64_t d, void* stream);

/* Backward: dq / dk / dv are written as bf16 column blocks (leading dimensions ld_dq / ld_dkv), no atomics.
 * workspace: mm355_attn_bwd_ws_floats(...) floats; NULL allowed when that is 0 (d == 128 always needs it: MM355_EINVAL without).
 * The d == 128 kernels (LLaMA-3) sum a GQA group in registers (the dK/dV workgroup walks all query heads of its KV group) and use the
 * workspace only for 2*B*Hq*L per-row constants; the generic-d kernels under GQA (e.g. TinyLlama, d = 64) run one workgroup per
 * (KV tile, query head) and need 2*B*L*Hq*d floats for the per-head partials that are summed afterwards. */
int mm355_attn_bwd(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k,
                   const mm355_bf16* d_o, int64_t ld_o, const float* lse, const float* delta, const int32_t* seqlens,
                   mm355_bf16* dq, int64_t ld_dq, mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv,
                   int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, float scale, int causal,
                   float* workspace, void* stream);

/* mm355_attn_bwd with the inverse RoPE rotation of dq / dk (what mm355_rope_qk(inverse = 1) does to the stored gradients: reference HF
 * apply_rotary_pos_emb backward, reached at metamorph_llama.py:349-359) applied in the kernels' epilogues: q and k are the POST-RoPE tensors the
 * forward attention saw, dq / dk come out as gradients of the PRE-RoPE projections.  Row l of sample b is position l (+ pos_offset[b]) of the
 * cos / sin tables ([positions][128] bf16).  d == 128 only (MM355_EUNSUPPORTED otherwise: call mm355_attn_bwd + mm355_rope_qk). */
int mm355_attn_bwd_rope(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k,
                        const mm355_bf16* d_o, int64_t ld_o, const float* lse, const float* delta, const int32_t* seqlens,
                        mm355_bf16* dq, int64_t ld_dq, mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv,
                        int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, float scale, int causal,
                        const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* pos_offset, float* workspace, void* stream);

/* The general form: mm355_attn_bwd (cos_t = sin_t = NULL) or mm355_attn_bwd_rope (tables given) with the kernel generation chosen by the
 * caller: variant 0 = the product's choice; for tests / tools 2 = the generic-d kernels (under GQA their 2*B*L*Hq*d workspace is the
 * caller's to provide), 3 = the round-2 two-waves-per-SIMD d == 128 kernels (only in -DMM355_LEGACY_VARIANTS builds, MM355_EUNSUPPORTED
 * otherwise), 4 = the one-wave-per-SIMD hand-placed streams (d == 128, workspace required), 41 = the same streams serialised
 * (bit-identical to 4 by construction).
 * Same reference call site as mm355_attn_bwd (backward of torch SDPA reached at metamorph_llama.py:349-359). */
int mm355_attn_bwd_variant(const mm355_bf16* q, const mm355_bf16* k, const mm355_bf16* v, int64_t ld_q, int64_t ld_k,
                           const mm355_bf16* d_o, int64_t ld_o, const float* lse, const float* delta, const int32_t* seqlens,
                           mm355_bf16* dq, int64_t ld_dq, mm355_bf16* dk, mm355_bf16* dv, int64_t ld_dkv,
                           int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, float scale, int causal,
                           const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* pos_offset,
                           float* workspace, int variant, void* stream);

/* floats of `workspace` mm355_attn_bwd needs for this geometry (ld_max = largest of ld_q / ld_k / ld_o); 0 when none is needed.
 * d == 128: 2 * B * Hq * L (the score chains' C operands -lse / scale and -delta); generic d under GQA: 2 * B * L * Hq * d. */
int64_t mm355_attn_bwd_ws_floats(int64_t B, int64_t L, int64_t Hq, int64_t Hkv, int64_t d, int64_t ld_max);

/* f32 [rows][cols] -> bf16 column block (generic helper). */
int mm355_cast_f32_bf16_2d(const float* in, int64_t ld_in, mm355_bf16* out, int64_t ld_out,
                           int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Decode shape (greedy_decode / generate with a KV cache; reference metamorph_llama.py:502-597, which
 * re-runs the prefix every step).  HBM-bound streaming kernels.
 *   gemv: y[M,N] = x[M,K] . W[N,K]^T for M <= 16 new rows -- and for 17 .. 32 rows on WIDE weights (more than 1280 groups of 16 weight rows, i.e.
 *         N > 20480 for gemv, I > 10240 for gemv_swiglu without norm_w: gate|up, lm_head), where a second group of 16 x rows rides on the same
 *         weight fragments -- else MM355_EUNSUPPORTED: use mm355_gemm_bf16 / mm355_gemm_splitk_bf16.  Up to four rows run on the
 *         vector ALU with the x rows parked in LDS (windows of 4096 columns), so that only weight loads sit in the in-order
 *         vector-memory queue: one or two rows as an fp32 fma chain, three and four on v_dot2c_f32_bf16; 5 .. 16 rows on
 *         v_mfma_f32_16x16x32_bf16, the weight rows loaded coalesced and re-laid out into fragments through wave-private LDS.  flags
 *         BIAS / GELU_ERF / GELU_TANH / RESIDUAL / OUT_F32 as for the GEMM.  The weight is addressed with 32-bit byte offsets:
 *         N * ldw * 2 < 3.75 GiB (else the plain stream of mm355_gemv_bf16 for up to four rows; MM355_EUNSUPPORTED for more rows and for
 *         the fused forms below).
 *   attn_decode: one query row per (sample, head), q [B][Hq*d] (ld_q), caches [B][max rows][Hkv*d] (row stride ld_kv,
 *         sample stride batch_stride_kv), kv_lens[B] (device) valid cached rows INCLUDING the current one, max_kv_len an
 *         upper bound of them (sizes the launch and the workspace of mm355_attn_decode_ws_floats floats); GQA groups 1/2/4/8.
 *         ONE launch: the key group that finishes last merges the partials of its heads; the arrival counters are the FIRST B*Hq
 *         words of the workspace (whatever max_kv_len) -- zero them once before the first call (hipMemset), every call leaves them zero.
 *   gemv_swiglu: act[M][I] = SiLU(g) * u with [g | u] = n . Wgu[2I][K]^T formed in the GEMV's epilogue (g, u rounded to bf16 first:
 *         the bits of mm355_gemv_bf16 + mm355_swiglu_fwd); norm_w != NULL: n = RMSNorm(x; norm_w, eps) formed per workgroup on the
 *         fly (the bits of mm355_rmsnorm_fwd), else n = x.  HF LlamaMLP / LlamaRMSNorm at decode shape (metamorph_llama.py:502-597).
 *         With norm_w and 5 .. 16 rows all normalised rows stay in LDS: M * (K rounded up to 32 + 8) * 2 bytes <= 140 KiB, else MM355_EUNSUPPORTED
 *         (run mm355_rmsnorm_fwd first); up to four rows any K (windows of 4096 columns).  The same holds for gemv_rope_append.
 *   gemv_rope_append: the fused q|k|v projection of M new rows with RoPE at positions[m] (device) and the KV-cache append in the
 *         epilogue: q -> qkv[m][0 .. Hq*d), rotated k and v -> cache row positions[m] (the bits of mm355_gemv_bf16 +
 *         mm355_rope_kv_append); the k | v columns of `qkv` are not written.  norm_w as above.
 * ------------------------------------------------------------------------------------------------ */
int mm355_gemv_bf16(const mm355_bf16* x, int64_t ldx, const mm355_bf16* W, int64_t ldw, void* y, int64_t ldy,
                    int64_t M, int64_t N, int64_t K, const mm355_bf16* bias, const mm355_bf16* residual, int64_t ldr,
                    uint32_t flags, void* stream);
 /* rope_kv_append: one new fused qkv row per sample [B][(Hq+2Hkv)*d]: q and k rotated at positions[b] (device), q left in
 *         place, rotated k and v written to cache row positions[b] (graph-replayable: no host-side position). */
int mm355_rope_kv_append(mm355_bf16* qkv, int64_t ld, int64_t B, int64_t Hq, int64_t Hkv, int64_t d,
                         const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* positions,
                         mm355_bf16* k_cache, mm355_bf16* v_cache, int64_t ld_kv, int64_t batch_stride_kv, void* stream);
int mm355_gemv_swiglu_bf16(const mm355_bf16* x, int64_t ldx, const mm355_bf16* Wgu, int64_t ldw, mm355_bf16* act, int64_t ld_act,
                           int64_t M, int64_t I, int64_t K, const mm355_bf16* norm_w, float eps, void* stream);
int mm355_gemv_rope_append_bf16(const mm355_bf16* x, int64_t ldx, const mm355_bf16* Wqkv, int64_t ldw, mm355_bf16* qkv, int64_t ld_qkv,
                                int64_t M, int64_t Hq, int64_t Hkv, int64_t d, int64_t K, const mm355_bf16* norm_w, float eps,
                                const mm355_bf16* cos_t, const mm355_bf16* sin_t, const int32_t* positions, mm355_bf16* k_cache,
                                mm355_bf16* v_cache, int64_t ld_kv, int64_t batch_stride_kv, void* stream);
int64_t mm355_attn_decode_ws_floats(int64_t B, int64_t Hq, int64_t d, int64_t max_kv_len);
int mm355_attn_decode(const mm355_bf16* q, int64_t ld_q, const mm355_bf16* k_cache, const mm355_bf16* v_cache,
                      int64_t ld_kv, int64_t batch_stride_kv, const int32_t* kv_lens, int64_t max_kv_len,
                      mm355_bf16* o, int64_t ld_o, int64_t B, int64_t Hq, int64_t Hkv, int64_t d, float scale,
                      float* workspace, void* stream);
/* tests / tools: variant 0 = as mm355_attn_decode (one 1024-thread workgroup per sample, KV head and 1024 cached rows; the lone
 * workgroup of a cache of <= 1024 rows writes the output itself: no device-scope fence -- and, when max_kv_len <= 1024, the GQA group is
 * spread over one workgroup per query head (pairs of heads from 64 workgroups on): same arithmetic, bit-identical output, half the
 * launch time), 1 = one 256-thread workgroup per 256-row chunk + merge by the last, 2 = variant 0 with the whole GQA group in one
 * workgroup whatever the bound.  max_kv_len is a LAUNCH parameter: a caller that knows its lengths on the host passes 1024 while every
 * kv_lens[b] <= 1024 and the capacity afterwards (metamorph_amd.functional.decode_kv_bound). */
int mm355_attn_decode_variant(const mm355_bf16* q, int64_t ld_q, const mm355_bf16* k_cache, const mm355_bf16* v_cache,
                      int64_t ld_kv, int64_t batch_stride_kv, const int32_t* kv_lens, int64_t max_kv_len,
                      mm355_bf16* o, int64_t ld_o, int64_t B, int64_t Hq, int64_t Hkv, int64_t d, float scale,
                      float* workspace, int variant, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise: SwiGLU (HF LlamaMLP; K12), GELU (projector / vision_head), scaling helpers.
 * gu = [M][2I] with gate in columns [0,I) and up in [I,2I).
 * ------------------------------------------------------------------------------------------------ */
int mm355_swiglu_fwd(const mm355_bf16* gu, mm355_bf16* act, int64_t M, int64_t I, void* stream);
int mm355_swiglu_bwd(const mm355_bf16* gu, const mm355_bf16* dact, mm355_bf16* dgu, mm355_bf16* act,
                     int64_t M, int64_t I, void* stream);
/* same, additionally emitting the contraction-major copies the weight-gradient GEMMs consume: actT[I][M] and dguT[2I][M]
 * (what mm355_transpose_bf16 would make of act and dgu).  M % 64 == 0 and I % 64 == 0, else MM355_EUNSUPPORTED. */
int mm355_swiglu_bwd_t(const mm355_bf16* gu, const mm355_bf16* dact, mm355_bf16* dgu, mm355_bf16* actT,
                       mm355_bf16* dguT, int64_t M, int64_t I, void* stream);
#define MM355_GELU_ERF  0
#define MM355_GELU_TANH 1
int mm355_gelu_fwd(const mm355_bf16* x, mm355_bf16* y, int64_t n, int kind, void* stream);
int mm355_gelu_bwd(const mm355_bf16* x, const mm355_bf16* dy, mm355_bf16* dx, int64_t n, int kind, void* stream);
/* x[i] *= *s_dev (optionally times s_host) */
int mm355_scale_bf16(mm355_bf16* x, int64_t n, const float* s_dev, float s_host, void* stream);
/* y[i] (+)= s * x[i]; y bf16, x bf16/f32 */
int mm355_axpy_bf16(mm355_bf16* y, const mm355_bf16* x, int64_t n, const float* s_dev, float s_host,
                    int accumulate, void* stream);
int mm355_axpy_f32_to_bf16(mm355_bf16* y, const float* x, int64_t n, float s_host, int accumulate, void* stream);
/* the same with a DEVICE scalar on top (s_dev nullable): y (+)= s_dev[0] * s_host * x -- fp32-accumulated lm_head weight gradient
 * scaled by the upstream loss gradient without a host sync (functional.LinearCrossEntropyFn) */
int mm355_axpy_f32_to_bf16_dev(mm355_bf16* y, const float* x, int64_t n, const float* s_dev, float s_host, int accumulate,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * Cross entropy over a chunk of rows of bf16 logits (metamorph_llama.py:398-413; K13).
 * logits [R][ld] bf16 (columns [V,ld) are padding).  targets[r] in [0,V) or < 0 = ignored row.
 * loss_sum += sum_r (lse_r - logit_r[target]);  then, IN PLACE, logits <- grad_scale*(softmax - onehot)
 * (zero on ignored rows and padding columns).  fp32 math on the bf16-rounded logits like the reference's
 * `.float()`; rows are compacted by the host so ignored rows normally never reach this kernel.
 * row_ws (nullable, R floats): the per-row NLL values are written there and added to loss_sum in a FIXED order by one workgroup
 * (bit-reproducible loss); NULL = one fp32 atomicAdd per row (order, hence the last bits, vary from run to run).  The same parameter
 * exists on the three image-AR loss entry points below.  loss_sum may be NULL when row_ws is given (per-row values only).
 * ------------------------------------------------------------------------------------------------ */
int mm355_ce_rows(mm355_bf16* logits, int64_t ld, const int32_t* targets, int64_t R, int64_t V,
                  float grad_scale, float* loss_sum, float* row_ws, void* stream);
/* out[0] = (accumulate ? out[0] : 0) + scale * sum(values[0..n)) in a fixed order (one workgroup; the reduction behind row_ws) */
int mm355_sum_rows_f32(const float* values, int64_t n, float scale, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * lm_head + shifted cross entropy, forward and both gradients, as ONE call (SURVEY 8(b) `linear_ce`; reference metamorph_llama.py:393-413:
 * `logits = lm_head(hidden).float()`, shift, `CrossEntropyLoss()` = mean NLL over the positions whose next label is live).
 *   hidden [*, ldh] bf16 rows of the final-norm output; rows[i] (int32, device; NULL = rows 0..n-1) = the row that predicts targets[i]
 *   (int32, device, in [0, V)) -- the host plan lists only positions with a live next label, so no flop is spent on ignored ones;
 *   W [V, ldw] = lm_head.weight.
 *   loss[0]        = (1/n) sum_i NLL_i, the per-row values summed in a fixed order (bit-reproducible);
 *   d_hidden [n,h] = d loss / d hidden[rows[i]] (compact, bf16; NULL = not wanted);
 *   dW [V,h]       = d loss / d W, OVERWRITTEN (bf16, or fp32 when dw_f32 != 0; NULL = not wanted).
 * The rows go through the logits GEMM in chunks of 8192: bf16 logits (the reference's bf16 nn.Linear output) -> mm355_ce_rows (fp32 math,
 * gradient in place) -> the two gradient GEMMs; the [n, V] fp32 logits tensor never exists.  Workspace: mm355_linear_ce_ws_bytes(n, V, h,
 * gather = rows != NULL, need_dh = d_hidden != NULL, need_dw = dW != NULL) bytes, 16-byte aligned.
 * ------------------------------------------------------------------------------------------------ */
int64_t mm355_linear_ce_ws_bytes(int64_t n, int64_t V, int64_t h, int gather, int need_dh, int need_dw);
int mm355_linear_ce(const mm355_bf16* hidden, int64_t ldh, const int32_t* rows, const int32_t* targets, int64_t n,
                    const mm355_bf16* W, int64_t ldw, int64_t V, int64_t h, float* loss, mm355_bf16* d_hidden, void* dW,
                    int dw_f32, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Splice (metamorph_arch.py:259-399; K6) driven by the host gather plan (bit-exact int bookkeeping):
 *   src[r] >= 0        -> row src[r] of embed_tokens.weight
 *   src[r] == -1       -> zero row (padding)
 *   src[r] <= -2       -> row (-2 - src[r]) of the projected image features
 * ------------------------------------------------------------------------------------------------ */
int mm355_splice_gather(const mm355_bf16* embed, const mm355_bf16* proj, const int32_t* src,
                        mm355_bf16* out, int64_t rows, int64_t h, void* stream);
/* backward into the projector output: dproj[n] = dout[row_of_feature[n]] or 0 when row_of_feature[n] < 0 */
int mm355_rows_gather(const mm355_bf16* in, int64_t ld_in, const int32_t* idx, mm355_bf16* out, int64_t ld_out,
                      int64_t R, int64_t h, void* stream);
/* dst[idx[r]] += src[r]   (idx unique) */
int mm355_rows_scatter_add(const mm355_bf16* src, int64_t ld_src, const int32_t* idx, mm355_bf16* dst,
                           int64_t ld_dst, int64_t R, int64_t h, void* stream);
/* embedding gradient: for segment s (token id tok[s]) sum rows pos[seg_start[s] .. seg_start[s+1]) of dout
 * in fp32 and (accumulate ? add to : store as) dembed[tok[s]].  Deterministic, no atomics. */
int mm355_embed_grad(const mm355_bf16* dout, const int32_t* tok, const int32_t* seg_start, const int32_t* pos,
                     int64_t n_seg, mm355_bf16* dembed, int64_t h, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Vision front/back ends.
 * im2col: images [N][3][H][W] (f32 or bf16) -> patches [N*gh*gw][Kp] bf16, k = c*p*p + dy*p + dx,
 *         columns [3*p*p, Kp) zero  (SigLIP patch embedding Conv2d 14x14/14, "valid"; K1).
 * bilinear_l2norm: [N][side_in^2][C] -> [N][side_out^2][C]; fp32 bilinear (align_corners=False), round to
 *         bf16, then F.normalize(p=2, eps 1e-12) in bf16 when normalize != 0 (siglip_encoder.py:151-163,206-208).
 * ------------------------------------------------------------------------------------------------ */
int mm355_im2col_patch(const void* images, int images_are_f32, int64_t N, int64_t H, int64_t W, int64_t p,
                       mm355_bf16* out, int64_t Kp, void* stream);
int mm355_bilinear_l2norm(const mm355_bf16* in, mm355_bf16* out, int64_t N, int64_t side_in, int64_t side_out,
                          int64_t C, int normalize, void* stream);
/* backward of the above for a trainable tower (freeze_vision=False, siglip_encoder.py:139): d_in_f32[N][side_in^2][C] +=
 * scatter of d/d(interpolated row) with the bilinear weights (caller zeroes d_in_f32; fp32 atomics). */
int mm355_bilinear_l2norm_bwd(const mm355_bf16* in, const mm355_bf16* d_out, float* d_in_f32, int64_t N, int64_t side_in,
                              int64_t side_out, int64_t C, int normalize, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Cosine regression loss (metamorph_llama.py:433-435,449-455; K16).  pred_raw = vision_head output.
 *   p = normalize ? F.normalize(pred_raw) (bf16 rounded) : pred_raw
 *   loss_sum += sum_r cos(target_r, p_r)  (caller turns it into -mean);  dpred = d(-mean cos)/d pred_raw * 1
 * ------------------------------------------------------------------------------------------------ */
int mm355_cosine_loss(const mm355_bf16* pred_raw, const mm355_bf16* target, int64_t R, int64_t C, int normalize,
                      float* cos_sum, mm355_bf16* dpred, float* row_ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The other two image-AR head variants (metamorph_llama.py:437-447 soft-CE, :459 + :211-219 mean-abs) and the
 * temperature softmax they pair with on the tower side (siglip_encoder.py:210-211) and in image-mode decoding
 * (metamorph_llama.py:372-373).  One wave per row, C % 8 == 0.
 *   mean_abs:  abs_sum[0] += sum |target - pred| (bf16 difference like the reference stack); caller divides by R*C.
 *              dpred (nullable) = d mean|t - p| / d pred.     (constructor default: normalize_vision=False, apply_softmax=False)
 *   soft_ce:   u = normalize ? F.normalize(pred_raw) : pred_raw;  q = softmax(u / temperature) (bf16);
 *              loss_sum[0] += -sum_r sum_j target[r][j] log(q[r][j] + 1e-10); caller divides by R.
 *              dpred (nullable) = d (loss_sum / R) / d pred_raw.
 *   softmax_rows:     y = softmax(x / temperature) per row (fp32 inside, bf16 out).
 *   softmax_rows_bwd: dx = y * (dy - sum_j dy_j y_j) / temperature.
 * ------------------------------------------------------------------------------------------------ */
int mm355_mean_abs_loss(const mm355_bf16* pred, const mm355_bf16* target, int64_t R, int64_t C, float* abs_sum,
                        mm355_bf16* dpred, float* row_ws, void* stream);
int mm355_soft_ce_loss(const mm355_bf16* pred_raw, const mm355_bf16* target, int64_t R, int64_t C, int normalize,
                       float temperature, float* loss_sum, mm355_bf16* dpred, float* row_ws, void* stream);
int mm355_softmax_rows(const mm355_bf16* x, mm355_bf16* y, int64_t R, int64_t C, float temperature, void* stream);
int mm355_softmax_rows_bwd(const mm355_bf16* y, const mm355_bf16* dy, mm355_bf16* dx, int64_t R, int64_t C,
                           float temperature, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ZeRO-2 shard update (replaces DeepSpeed zero2.json + HF adamw_torch, train.py:82): AdamW on the
 * rank's fp32 master shard, writes the updated bf16 parameters.  grad_scale_dev (nullable) is a device
 * scalar multiplied into the gradient (1/world, clip coefficient).
 * ------------------------------------------------------------------------------------------------ */
int mm355_adamw_shard(float* p32, float* m, float* v, const mm355_bf16* g, mm355_bf16* p_out, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      float bias_corr1, float bias_corr2, const float* grad_scale_dev, void* stream);
/* out[0] += sum x^2 (grad-norm partial).  Deterministic (no atomics): per-workgroup sums go to `partials` (caller-allocated,
 * MM355_SUMSQ_PARTIALS floats, contents scratch) and are added in index order by a second one-workgroup launch. */
#define MM355_SUMSQ_PARTIALS 2048
int mm355_sumsq_bf16(const mm355_bf16* x, int64_t n, float* out, float* partials, void* stream);
/* clip coefficient: coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) * pre_scale */
int mm355_clip_coef(const float* sumsq, float max_norm, float pre_scale, float* coef, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MM355_H */
