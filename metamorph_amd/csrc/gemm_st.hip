// bf16 MFMA GEMM for gfx950, one wave per SIMD:  C[M,N] = epilogue(A[M,K] . B[N,K]^T), variant 13 of mm355_gemm_bf16.
//
// A persistent workgroup of four waves (one per SIMD, the whole 512-entry register file each) walks 256 x 256 output tiles; a wave owns
// a 128 x 128 wave tile as 8 x 8 blocks of v_mfma_f32_16x16x32_bf16 with all 256 accumulator registers in a[0:255].  Against the
// eight-wave ping-pong kernel (gemm_pp_tile: 128 x 64 wave tiles, two waves per SIMD) a K tile costs 32 instead of 48 fragment
// ds_read_b128 per SIMD, there is no hand-over between wave groups, and the next tile's first two K stages are already in flight (LDS-DMA)
// while a tile's accumulators drain.  The main loop is a hand-placed instruction stream (tools/gen_gemm_st.py -> gemm_st_gen/*.inc, one
// asm statement per instruction on literal registers); hipcc owns v[0:95] only (amdgpu_num_vgpr(96), checked by tools/audit_attn4.py).
//
// LDS (160 KiB): two 64-KiB K stages [X 256 rows | W 256 rows] x 128 B (16-B chunk ^ (row & 7), filled lane-linearly by LDS-DMA with
// the swizzle on the source chunk) + 32 KiB of fp32 epilogue staging (8 KiB per wave), so the ring keeps filling during an epilogue.
// Rows past M / N are out of range of the tile's buffer descriptor (zeros, no access); K % 128 == 0, K >= 256.
// Summation order over k = gemm_pp_tile's: the outputs are bit-identical to variant 11 (tests/test_kernels_gpu.py).
#include "gemm_common.h"

#ifndef GST_GEN_DIR
#define GST_GEN_DIR gemm_st_gen
#endif
#define GST_STR2(x) #x
#define GST_STR(x) GST_STR2(x)
#define GST_INC(f) GST_STR(GST_GEN_DIR/f)

namespace {

constexpr int ST_STAGE = 65536;                              // one K stage: X image then W image
constexpr int ST_STG = 2 * ST_STAGE;                         // epilogue staging
constexpr int ST_LDS = ST_STG + 4 * 8192;                    // 163 840 B

#define GST_BARRIER() asm volatile("s_barrier" ::: "memory")
// piece i of this wave (i < 8: X rows 8 (wave + 4 i) .., i >= 8: W rows) of the stage the fetch state points at into ring slot p:
// M0 = LDS destination (the dynamic LDS starts at address 0: wave * 1024 + a literal), lane-linear 16 B per lane
#define GST_DMA_ASM(i, p) asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" \
    :: "s"(wbase), "i"((p) * ST_STAGE + (i) * 4096), "v"(vo[i]), "s"((i) < 8 ? rsX : rsW), "s"(GST_KOFF(koff)) : "memory", "scc")
#ifdef GST_DMA_W0                                            // TIMING ONLY: wave 0 alone issues its pieces
#define GST_DMA(i, p) do { if (wave == 0) GST_DMA_ASM(i, p); } while (0)
#else
#define GST_DMA(i, p) GST_DMA_ASM(i, p)
#endif
#define GST_DMA_IF(i, p, k) do { if (wave == (k)) GST_DMA_ASM(i, p); } while (0)
#ifdef GST_L2RES                                             // TIMING ONLY: every stage fetched from K stages 0 / 1 (L2-resident sources)
#define GST_KOFF(k) ((k) & 128)
#else
#define GST_KOFF(k) (k)
#endif
#define GST_NEXT_STAGE() do { koff += 128; } while (0)

template <bool SAFE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(96))) void gemm_st_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("" ::: "v255", "a255");                     // the stream's literal registers: the descriptor must allocate the whole file
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int M = a.M, N = a.N;
    const int nk = a.K >> 6;                                 // host: K % 128 == 0, K >= 256

    // ---- tile walk: workgroup b runs on XCD b & 7; XCD x owns a contiguous range of the (grouped-raster) tile order and its
    //      workgroups take it round-robin, so that the tiles in flight on one L2 are neighbours
    const int total = a.ntm * a.ntn;
    const int q8 = total >> 3, r8 = total & 7;
    const int xcd = blockIdx.x & 7, per_x = gridDim.x >> 3;
    const int x_count = q8 + (xcd < r8 ? 1 : 0);
    const int x_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int GM = a.gm, gsize = GM * a.ntn;
    auto coords = [&](int idx, int& m0, int& n0) {
        const int logical = x_start + idx;
        const int grp = logical / gsize;
        const int first_m = grp * GM;
        const int gm = min(a.ntm - first_m, GM);
        const int in_g = logical - grp * gsize;
        m0 = (first_m + in_g % gm) * 256;
        n0 = (in_g / gm) * 256;
    };
    int idx = blockIdx.x >> 3;
    if (idx >= x_count) return;

    // ---- fragment read addresses: [slot][sub-step]; row = lane & 15 (+ 16 per block: immediate), chunk = (4 u + lane >> 4) ^ (row & 7)
    uint32_t XA[2][2], WA[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t sw = (uint32_t)(((4 * u + fq) ^ (fr & 7)) << 4);
            XA[p][u] = p * ST_STAGE + (wm * 128 + fr) * 128 + sw;
            WA[p][u] = p * ST_STAGE + 32768 + (wn * 128 + fr) * 128 + sw;
            asm volatile("" : "+v"(XA[p][u]));
            asm volatile("" : "+v"(WA[p][u]));
        }
    // ---- LDS-DMA source offsets from the tile's first row: piece = 8 rows x 128 B, lane -> (row lane >> 3, LDS chunk lane & 7)
    uint32_t vo[16];
    {
#ifdef GST_NOSWZ                                             // TIMING ONLY: natural chunk order on the source side
        const int rin = lane >> 3, c = (lane & 7);
#else
        const int rin = lane >> 3, c = (lane & 7) ^ rin;
#endif
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (wave + 4 * (i & 7)) * 8 + rin;
            vo[i] = (uint32_t)row * (uint32_t)((i < 8 ? a.lda : a.ldb) * 2) + (uint32_t)(c * 16);
            asm volatile("" : "+v"(vo[i]));
        }
    }
    // staging: write address of accumulator block jn = 2 j + b (+ 128 j): row m = lane & 15, chunk (4 jn + lane >> 4) ^ (m & 7)
    uint32_t SW[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
        SW[b] = ST_STG + wave * 8192 + fr * 512 + ((((b ^ ((fr >> 2) & 1)) << 2) + (fq ^ (fr & 3))) << 4);

    int m0, n0;
    coords(idx, m0, n0);
    // descriptors: base = the tile's first row, num_records = the bytes up to the end of its last valid row (a row's 128-B piece of the
    // current K stage is addressed by the scalar offset, which the range check ignores: the check is on row * ld * 2 + chunk only)
    auto rsrc = [](const uint16_t* ptr, int64_t ld, int r0, int rows) -> u32x4 {   // raw buffer: 48-bit base, stride 0, num_records in bytes
        const uint64_t p = (uint64_t)(ptr + (int64_t)r0 * ld);
        return u32x4{(uint32_t)p, (uint32_t)(p >> 32) & 0xffffu, (uint32_t)min(rows - r0, 256) * (uint32_t)(ld * 2), 0x00020000u};
    };
    u32x4 rsX = rsrc(a.A, a.lda, m0, M), rsW = rsrc(a.B, a.ldb, n0, N);
    int koff = 0;
    const uint32_t wbase = (uint32_t)wave * 1024u;

    // ---- prologue: stages 0 and 1 of the first tile, fragments of (0, 0)
#pragma unroll
    for (int i = 0; i < 16; ++i) GST_DMA(i, 0);
    GST_NEXT_STAGE();
#pragma unroll
    for (int i = 0; i < 16; ++i) GST_DMA(i, 1);
    GST_NEXT_STAGE();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    GST_BARRIER();
#include GST_INC(pro_reads.inc)

    const uint32_t fl = a.flags;
    const bool vec_ok = ((a.ldc & 7) == 0) && (!(fl & MM355_GEMM_RESIDUAL) || (a.ldr & 7) == 0);
    for (;;) {
        // the next tile of this workgroup (none: empty descriptors -- every lane out of range, zeros into the ring, no memory access)
        const int idx_n = idx + per_x;
        const bool has_next = idx_n < x_count;
        int m0n = 0, n0n = 0;
        if (has_next) coords(idx_n, m0n, n0n);
        // stage t fetches stage t + 2: the first trip stages 2, 3 of this tile (koff is already 256)
        if constexpr (SAFE) {
#include GST_INC(safe_trip_first.inc)
        } else {
#include GST_INC(trip_first.inc)
        }
        GST_NEXT_STAGE();
        for (int t = 2; t < nk - 2; t += 2) {
            if constexpr (SAFE) {
#include GST_INC(safe_trip.inc)
            } else {
#include GST_INC(trip.inc)
            }
            GST_NEXT_STAGE();
        }
        // last trip: stages 0, 1 of the next tile
        rsX = rsrc(a.A, a.lda, m0n, has_next ? M : m0n);
        rsW = rsrc(a.B, a.ldb, n0n, has_next ? N : n0n);
        koff = 0;
        if constexpr (SAFE) {
#include GST_INC(safe_trip.inc)
        } else {
#include GST_INC(trip.inc)
        }
        GST_NEXT_STAGE();
        // ---- epilogue: 8 m blocks of 16 rows, each accumulators -> this wave's fp32 slab -> row-contiguous fused stores
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results
#pragma unroll 1
        for (int im_ = 0; im_ < 8; ++im_) {
#include GST_INC(drain.inc)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned char* slab = smem + ST_STG + wave * 8192;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 4 + fq;                 // 16 lanes per row, lane & 15 = group of eight columns
                const int grow = m0 + wm * 128 + im_ * 16 + row;
                const int c = n0 + wn * 128 + fr * 8;
                const f32x4 s0 = *(const f32x4*)(slab + row * 512 + (((2 * fr) ^ (row & 7)) << 4));
                const f32x4 s1 = *(const f32x4*)(slab + row * 512 + (((2 * fr + 1) ^ (row & 7)) << 4));
                if (grow < M && c < N) {
                    float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    const int64_t rr = (fl & MM355_GEMM_RESIDUAL) ? (a.res_mod > 0 ? (int64_t)(grow % a.res_mod) : (int64_t)grow) : 0;
                    epi_store8(a, fl, vec_ok, grow, rr, c, v);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab is rewritten by the next block's drain
        }
        if (!has_next) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stores and loads retire out of order: restart the counted protocol clean
        idx = idx_n;
        m0 = m0n;
        n0 = n0n;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace

// launcher used by mm355_gemm_bf16 (gemm_bf16.hip): variant 13 (placed stream) / 14 (serialised stream: the hazard detector)
int mm355_gemm_st_launch(const void* args, int serialised, void* stream) {
    GemmArgs a = *(const GemmArgs*)args;
    a.ntm = (a.M + 255) / 256;
    a.ntn = (a.N + 255) / 256;
    a.gm = 4;
    const int64_t total = (int64_t)a.ntm * a.ntn;
    if (total <= 0 || total > 0x7fffffff) return MM355_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return MM355_ELAUNCH;
    int grid = (cus / 8) * 8;                                // one workgroup per CU, a whole number per XCD
    if (grid < 8) grid = 8;
    const int64_t need = ((total + 7) / 8) * 8;              // never more workgroups than tiles (per XCD)
    if (grid > need) grid = (int)need;
    static std::atomic<uint64_t> ok0{0}, ok1{0};
    if (serialised) {
        if (mm_ensure_dynamic_lds((const void*)gemm_st_kernel<true>, ST_LDS, ok1) != MM355_OK) return MM355_ELAUNCH;
        hipLaunchKernelGGL(gemm_st_kernel<true>, dim3((unsigned)grid), dim3(256), ST_LDS, (hipStream_t)stream, a);
    } else {
        if (mm_ensure_dynamic_lds((const void*)gemm_st_kernel<false>, ST_LDS, ok0) != MM355_OK) return MM355_ELAUNCH;
        hipLaunchKernelGGL(gemm_st_kernel<false>, dim3((unsigned)grid), dim3(256), ST_LDS, (hipStream_t)stream, a);
    }
    return mm_launch_status();
}
