// Attention, d == 128: the 64-query-rows-per-wave instantiations of the forward / dQ kernel templates (attn3_kernels.h).
// One workgroup of four waves per CU = one wave per SIMD with the whole 512-entry register file (accumulators in AGPRs): this file is
// built WITHOUT -mllvm -amdgpu-mfma-vgpr-form=1 (metamorph_amd/build.py), which the rest of the library uses to keep its two-waves-per-
// SIMD kernels inside 256 VGPRs.  Why 64 rows: every K / V fragment read (ds_read_b128 rows, ds_read_b64_tr_b16 gathers) then feeds
// four MFMAs instead of two.  (The premise did not hold up: with eight waves reading, the gathers cost 2.4 LDS cycles and the row reads
// 4.3, the LDS is 20 % busy in the 32-row kernels -- tools/probes/lds_throughput_probe.hip)
// EXPERIMENT, opt-in (MM355_ATTN_RQ=4 / MM355_ATTN_RQ_DQ=4): parity-green but slower than the 32-row default, see attn3_kernels.h.
#include "attn3_kernels.h"

int mm355_attn3_fwd_wide_launch(const attn2::Args& a, hipStream_t s) {
    const int64_t nblk = (int64_t)((a.L + 255) / 256) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(attn3::fwd_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return mm_launch_status();
}

int mm355_attn3_dq_wide_launch(const attn2::Args& a, hipStream_t s) {
    const int64_t nblk = (int64_t)((a.L + 255) / 256) * a.Hq * a.B;
    if (nblk > 0x7fffffff) return MM355_EINVAL;
    hipLaunchKernelGGL(attn3::dq_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return mm_launch_status();
}
