// mm355_linear_ce: lm_head + shifted cross entropy over the rows that carry a target, forward AND both gradients, as ONE C entry point
// (SURVEY.md 8(b) op list: linear_ce; reference metamorph_llama.py:393-413: `logits = lm_head(hidden).float()`, shift, CrossEntropyLoss).
//
// Host-side orchestration only -- every arithmetic step is an existing kernel of this library, launched on the caller's stream:
//   rows_gather  ->  per chunk of <= 8192 rows { logits GEMM (bf16 out, like the reference's bf16 nn.Linear) -> ce_rows (fp32 softmax / NLL on the
//   bf16-rounded logits, gradient written in place, per-row NLL values kept) -> dX GEMM -> (transposes) dW GEMM }  ->  fixed-order sum.
// The [n, V] fp32 logits tensor of the reference never exists; one chunk's bf16 logits live in the workspace.
#include "mm355_common.h"

#include <algorithm>

namespace {

constexpr int64_t CE_CHUNK = 8192;

inline int64_t up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
// contraction length a transposed copy of R rows gets: whole pairs of 64-wide K tiles for the ping-pong kernel once R is long, else a legal K
inline int64_t padded_rows(int64_t R) { return R >= 512 ? up(R, 128) : up(R, 8); }

struct Layout {
    int64_t Vp, chunk, rp, hc, wt, logits, dyT, xT, rows, total;
};

Layout layout(int64_t n, int64_t V, int64_t h, int gather, int need_dh, int need_dw) {
    Layout l{};
    l.Vp = up(V, 128);
    l.chunk = std::min(n, CE_CHUNK);
    l.rp = padded_rows(l.chunk);
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += up(bytes, 256); return o; };
    l.hc = gather ? take(n * h * 2) : -1;
    l.wt = need_dh ? take(h * l.Vp * 2) : -1;
    l.logits = take(l.chunk * l.Vp * 2);
    l.dyT = need_dw ? take(V * l.rp * 2) : -1;
    l.xT = need_dw ? take(h * l.rp * 2) : -1;
    l.rows = take(n * 4);
    l.total = off;
    return l;
}

}  // namespace

extern "C" int64_t mm355_linear_ce_ws_bytes(int64_t n, int64_t V, int64_t h, int gather, int need_dh, int need_dw) {
    if (n <= 0 || V <= 0 || h <= 0) return 0;
    return layout(n, V, h, gather, need_dh, need_dw).total;
}

extern "C" int mm355_linear_ce(const mm355_bf16* hidden, int64_t ldh, const int32_t* rows, const int32_t* targets, int64_t n,
                               const mm355_bf16* W, int64_t ldw, int64_t V, int64_t h, float* loss, mm355_bf16* d_hidden, void* dW,
                               int dw_f32, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!hidden || !targets || !W || !loss || !workspace || n <= 0 || V <= 0 || h <= 0 || (h & 7) || (ldh & 7) || (ldw & 7) || ldh < h || ldw < h ||
        n > 0x7fffffff || !mm_aligned16(workspace))
        return MM355_EINVAL;
    const int need_dh = d_hidden != nullptr, need_dw = dW != nullptr, gather = rows != nullptr;
    const Layout l = layout(n, V, h, gather, need_dh, need_dw);
    if (workspace_bytes < l.total) return MM355_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const mm355_bf16* hc = hidden;
    int64_t ldc = ldh;
    int rc;
    if (gather) {                                            // compact the rows that carry a target: no flop is spent on ignored positions
        mm355_bf16* g = (mm355_bf16*)(ws + l.hc);
        if ((rc = mm355_rows_gather(hidden, ldh, rows, g, h, n, h, stream)) != MM355_OK) return rc;
        hc = g;
        ldc = h;
    }
    mm355_bf16* wt = need_dh ? (mm355_bf16*)(ws + l.wt) : nullptr;
    if (need_dh) {                                           // W^T [h, Vp]: the B operand of the dX GEMM (K = Vp; padding columns zero)
        if (l.Vp != V && hipMemset2DAsync(wt + V, l.Vp * 2, 0, (l.Vp - V) * 2, h, s) != hipSuccess) return MM355_ELAUNCH;
        if ((rc = mm355_transpose_bf16(W, ldw, V, h, wt, l.Vp, stream)) != MM355_OK) return rc;
    }
    mm355_bf16* logits = (mm355_bf16*)(ws + l.logits);
    float* row_nll = (float*)(ws + l.rows);
    const float inv = 1.0f / (float)n;
    bool first = true;
    for (int64_t r0 = 0; r0 < n; r0 += CE_CHUNK) {
        const int64_t r = std::min(n - r0, CE_CHUNK);
        const mm355_bf16* x = hc + r0 * ldc;
        if ((rc = mm355_gemm_bf16(x, ldc, W, ldw, logits, l.Vp, r, V, h, nullptr, nullptr, 0, 0, 0u, 0, stream)) != MM355_OK) return rc;
        // logits <- d loss / d logits = inv * (softmax - onehot), padding columns 0; row r's NLL -> row_nll[r0 + r]
        if ((rc = mm355_ce_rows(logits, l.Vp, targets + r0, r, V, inv, nullptr, row_nll + r0, stream)) != MM355_OK) return rc;
        if (need_dh &&
            (rc = mm355_gemm_bf16(logits, l.Vp, wt, l.Vp, d_hidden + r0 * h, h, r, h, l.Vp, nullptr, nullptr, 0, 0, 0u, 0, stream)) != MM355_OK)
            return rc;
        if (need_dw) {                                       // dW[V, h] (+)= dlogits^T . x, contraction over the chunk's rows
            const int64_t rp = padded_rows(r);
            mm355_bf16* dyT = (mm355_bf16*)(ws + l.dyT);
            mm355_bf16* xT = (mm355_bf16*)(ws + l.xT);
            if (rp != r) {
                if (hipMemset2DAsync(dyT + r, rp * 2, 0, (rp - r) * 2, V, s) != hipSuccess) return MM355_ELAUNCH;
                if (hipMemset2DAsync(xT + r, rp * 2, 0, (rp - r) * 2, h, s) != hipSuccess) return MM355_ELAUNCH;
            }
            if ((rc = mm355_transpose_bf16(logits, l.Vp, r, V, dyT, rp, stream)) != MM355_OK) return rc;
            if ((rc = mm355_transpose_bf16(x, ldc, r, h, xT, rp, stream)) != MM355_OK) return rc;
            const uint32_t flags = (first ? 0u : MM355_GEMM_ACCUMULATE) | (dw_f32 ? MM355_GEMM_OUT_F32 : 0u);
            if ((rc = mm355_gemm_bf16(dyT, rp, xT, rp, dW, h, V, h, rp, nullptr, nullptr, 0, 0, flags, 0, stream)) != MM355_OK) return rc;
        }
        first = false;
    }
    // mean NLL over the n rows, summed in a fixed order (independent of the chunking): the loss scalar is bit-reproducible
    return mm355_sum_rows_f32(row_nll, n, inv, loss, 0, stream);
}
