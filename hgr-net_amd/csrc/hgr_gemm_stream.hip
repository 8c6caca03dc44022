// gemm_nt_stream: C[M,N] fp32 = A[M,K] . W[N,K]^T for FEW rows and MANY columns - the image-vs-class logits of
// tree_model.forward (model/clip_tree.py:331-333: feats @ zsl_weights.T; M = batch 512, N = 21 841 classes, K = 512).
//
// Why a second GEMM kernel.  This product is co-bound by HBM (44.7 MB of fp32 logits written, 22.4 MB of class embeddings
// read) and by latency: with K = 512 a tile's main loop is 8 K-tiles, so the tiled kernels of hgr_gemm.hip spend most of a
// tile's life in its prologue / epilogue and fetch one K-tile ahead (23-29 us, 16-20 % of the MFMA peak, 2.3-2.9 TB/s).
// Here a workgroup is PERSISTENT and the three jobs of a GEMM are given to different waves, so that none of them ever
// waits for another one's memory traffic (vmcnt retires in issue order inside a wave: a wave that both stores C and waits
// for operand loads waits for its stores too - DESIGN.md section 4.1, finding 3):
//   * waves 4-5 ("loaders") only issue LDS-DMA (global_load_lds_dwordx4): a ring of 3 stages of 40 KB, each stage one
//     64-deep K-tile of the workgroup's 64 A rows (8 KB) and 256 W rows (32 KB); two stages (80 KB per CU) are always in
//     flight, across K-tiles AND across output tiles - the stream of class embeddings never stops at a tile boundary;
//   * waves 0-3 ("MFMA waves") own 64 (M) x 64 (N) of the 64 x 256 output tile each: 4 x 4 tiles of
//     v_mfma_f32_16x16x32, fragments by ds_read_b128 from the swizzled stage (same layout as gemm_nt_128), and at the end
//     of a tile they fire its 16 stores per lane and walk on - they never execute a vmcnt wait;
//   * one raw s_barrier per K-tile: "stage t has landed" (loaders waited for their own DMA with a counted vmcnt) and
//     "stage t-1 has been read" (the slot the loaders refill next).
// W is the MFMA A operand and the activations the B operand (as in gemm_nt_128), so a lane's 4 accumulator registers are
// 4 consecutive n of one row m: 16-byte stores, 64 contiguous bytes per row per instruction, 256 per row per tile.
// The K reduction runs in the same order as in gemm_nt_128 / gemm_nt_256, so the results are bit-identical to theirs.
//
// Work split: ceil(M / 64) row panels x `nstreams` column ranges = one workgroup each (256 = one per CU at M = 512); a
// range is a whole number of 16-column groups, walked in steps of 256 columns.  The panels of one range sit on one XCD
// (blockIdx & 7) and run in lock-step by construction, so a class embedding is fetched from HBM once and hit in that
// XCD's L2 by the other panels.
#include "hgr_common.h"
#include <type_traits>

namespace {

constexpr int SM = 64, SN = 256, SK = 64, NSTAGE = 3;
constexpr int STAGE = (SM + SN) * SK * 2;          // 40 960 bytes
constexpr int PIECES = STAGE / 1024;               // 40 LDS-DMA instructions (1 KB = 8 rows of 128 B) per stage
constexpr int NLOADER = 2, PER_LOADER = PIECES / NLOADER;
constexpr int NTHREADS = (4 + NLOADER) * 64;

struct StreamArgs {
    const char *A; int64_t lda;
    const char *W; int64_t ldw;
    float *C; int64_t ldc;
    int M, N, K;
    int panels_m;      // ceil(M / 64)
    int per;           // columns per range (a multiple of 16)
    int xcd_map;       // 1: gridDim.x % 8 == 0, consecutive linear ids are dealt to one XCD
};

#define HGR_SBAR() do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int DT>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_stream(StreamArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lin = blockIdx.x;
    if (p.xcd_map) lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int panel = lin % p.panels_m, range = lin / p.panels_m;
    const int m0 = panel * SM;
    const int nbeg = range * p.per, nend = min(p.N, nbeg + p.per);
    const int nsteps = nbeg < nend ? (nend - nbeg + SN - 1) / SN : 0;
    const int nk = p.K / SK;
    const int total = nsteps * nk;                 // K-tiles this workgroup streams; identical for all of its waves
    if (total == 0) return;

    if (wave >= 4) {
        // ---- loader waves: the only issuers of operand loads -----------------------------------------------------
        // Loader 0 moves the 8 A pieces and the first 12 W pieces of a stage, loader 1 the other 20 W pieces.  Per-lane
        // source pointers are kept in registers (A: fixed for the whole kernel; W: re-derived once per 256-column
        // step), so an issue is one 64-bit add per instruction - no argument reloads, no branches in the stream.
        const int lrow = lane >> 3, c = (lane & 7) ^ lrow;      // row inside the 8-row piece; swizzled source chunk
        const char *Ab = p.A, *Wb = p.W;
        const int64_t lda2 = p.lda * 2, ldw2 = p.ldw * 2;
        const int Mm1 = p.M - 1, Nm1 = p.N - 1;
        auto loader = [&](auto which) {
            constexpr int L = decltype(which)::value;
            constexpr int NA = L == 0 ? SM / 8 : 0;             // A pieces of this loader
            constexpr int NW = PER_LOADER - NA;                 // W pieces
            constexpr int W0 = L == 0 ? 0 : PER_LOADER - SM / 8;  // first W piece (in 8-row units inside the 256-row tile)
            const char *asrc[NA ? NA : 1], *wsrc[NW];
#pragma unroll
            for (int q = 0; q < NA; ++q) asrc[q] = Ab + (int64_t)min(m0 + q * 8 + lrow, Mm1) * lda2 + c * 16;
            auto point = [&](int step) {
                const int n0 = nbeg + step * SN;
#pragma unroll
                for (int q = 0; q < NW; ++q) wsrc[q] = Wb + (int64_t)min(n0 + (W0 + q) * 8 + lrow, Nm1) * ldw2 + c * 16;
            };
            int is_step = 0, is_kt = 0, is_slot = 0;            // position of the next stage to issue
            point(0);
            auto issue = [&]() {
                char *dst = smem + is_slot * STAGE + L * PER_LOADER * 1024;
                const int koff = is_kt * (SK * 2);
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    __builtin_amdgcn_global_load_lds((const AS1 void *)(asrc[q] + koff), (AS3 void *)(dst + q * 1024), 16, 0, 0);
#pragma unroll
                for (int q = 0; q < NW; ++q)
                    __builtin_amdgcn_global_load_lds((const AS1 void *)(wsrc[q] + koff), (AS3 void *)(dst + (NA + q) * 1024), 16, 0, 0);
                if (++is_kt == nk) { is_kt = 0; ++is_step; point(is_step); }
                if (++is_slot == NSTAGE) is_slot = 0;
            };
            issue();
            if (total > 1) issue();
            for (int t = 0; t < total; ++t) {
                // my share of stage t has landed; stage t+1 (my PER_LOADER youngest instructions) may stay in flight
                if (t + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_LOADER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                HGR_SBAR();                                      // barrier t
                if (t + 2 < total) issue();                      // into the slot of stage t-1: every wave has read it
            }
        };
        if (wave == 4) loader(std::integral_constant<int, 0>());
        else loader(std::integral_constant<int, 1>());
        return;
    }

    // ---- MFMA waves ----------------------------------------------------------------------------------------------
    const int r = lane & 15, g = lane >> 4;
    const int offA = r * 128;                                    // + m tile * 2048
    const int offW = SM * 128 + (wave * 64 + r) * 128;           // + n tile * 2048
    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;
    const bool rows_full = m0 + SM <= p.M;
    int slot = 0;
    for (int step = 0; step < nsteps; ++step) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            HGR_SBAR();                                          // stage landed, visible to every wave
            const char *st = smem + slot * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int sw = kk ? sw1 : sw0;
                vec8 wf[4], xf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    wf[t] = *(const vec8 *)(st + offW + t * 2048 + sw);
                    xf[t] = *(const vec8 *)(st + offA + t * 2048 + sw);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = T16<DT>::mfma16(wf[i], xf[j], acc[i][j]);
            }
            if (++slot == NSTAGE) slot = 0;
        }
        // tile (i, j) of this lane: C[m0 + j*16 + r][nw + i*16 + g*4 .. +3]
        const int nw = nbeg + step * SN + wave * 64;
        if (rows_full && nw + 64 <= nend) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float *row = p.C + (int64_t)(m0 + j * 16 + r) * p.ldc + nw + g * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4 *)(row + i * 16) = acc[i][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + j * 16 + r;
                if (m >= p.M) continue;
                float *row = p.C + (int64_t)m * p.ldc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = nw + i * 16 + g * 4;
                    if (n + 3 < nend) *(f32x4 *)(row + n) = acc[i][j];
                    else
                        for (int e = 0; e < 4; ++e) if (n + e < nend) row[n + e] = acc[i][j][e];
                }
            }
        }
    }
}

}  // namespace

// Called by hgr_gemm_nt (hgr_gemm.hip) for fp32, epilogue-free products with few rows and many columns; returns 1 when
// it launched, 0 when the shape is not its kind (the caller then takes the tiled kernels).  Arguments were validated
// by the caller: K % 64 == 0, lda / ldw multiples of 8, 16-byte aligned A / W.
int hgr_gemm_stream_launch(const void *A, int64_t lda, const void *W, int64_t ldw, float *C, int64_t ldc,
                           int M, int N, int K, int dtype, void *stream) {
    if (M > 1024 || N < 4096 || K < 128) return 0;
    if ((ldc & 3) || !hgr_aligned(C, 16)) return 0;
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus = n;
    }
    StreamArgs a;
    a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    a.panels_m = (M + SM - 1) / SM;
    const int nstreams = cus / a.panels_m > 0 ? cus / a.panels_m : 1;
    const int groups = (N + 15) / 16;
    a.per = ((groups + nstreams - 1) / nstreams) * 16;
    const int grid = nstreams * a.panels_m;
    a.xcd_map = (grid % 8 == 0) ? 1 : 0;
    if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_stream<HGR_BF16>), dim3(grid), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_nt_stream<HGR_F16>), dim3(grid), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    return 1;
}
