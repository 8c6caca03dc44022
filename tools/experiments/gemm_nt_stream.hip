// EXPERIMENT - NOT part of libhgr.so (kept for the next round; results in tools/experiments/README.md and profiles/NOTES.md 4.1b).
// Two persistent kernels for the class-logits product that were built, validated bit-exact / exact on MI355X, measured,
// and NOT adopted: in the evaluation step (operands cold in HBM) the tiled kernels of hgr_gemm.hip are faster.
//   gemm_nt_stream  LDS ring fed by loader waves, MFMA waves never wait on memory      22.0 us warm / 32.6 us in the step
//   gemm_nt_wres    W slice resident in the LDS, A streamed straight into registers    30.3 us warm / 34.4 us in the step
//   (tiled gemm_nt_256 / gemm_nt_128 of libhgr:                                         25.5 us warm / 28.3 us in the step)
// Build as a variant library (see README.md): compile this file, link it with csrc/build/*.o, and add the three-line hook
// in hgr_gemm_nt that calls hgr_gemm_stream_launch() for epilogue-free fp32 products (HGR_GEMM_STREAM=1|2 picks the kernel).
//
// gemm_nt_stream: C[M,N] fp32 = A[M,K] . W[N,K]^T for FEW rows and MANY columns - the image-vs-class logits of
// tree_model.forward (model/clip_tree.py:331-333: feats @ zsl_weights.T; M = batch 512, N = 21 841 classes, K = 512).
//
// Why a second GEMM kernel.  This product is co-bound by HBM (44.7 MB of fp32 logits written, 22.4 MB of class embeddings
// read) and by latency: with K = 512 a tile's main loop is 8 K-tiles, so the tiled kernels of hgr_gemm.hip spend most of a
// tile's life in its prologue / epilogue and fetch one K-tile ahead (23-29 us, 16-20 % of the MFMA peak, 2.3-2.9 TB/s).
// Here a workgroup is PERSISTENT and the three jobs of a GEMM are given to different waves, so that none of them ever
// waits for another one's memory traffic (vmcnt retires in issue order inside a wave: a wave that both stores C and waits
// for operand loads waits for its stores too - profiles/NOTES.md section 4.1, finding 3):
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
#include "../../hgr-net_amd/csrc/hgr_common.h"
#include <stdlib.h>
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

// the logits are written once and read by the next kernel at the earliest: non-temporal stores keep them from displacing
// the class embeddings the other row panels are about to hit in the L2 (23.0 -> 22.5 us)
#define HGR_STREAM_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))

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
                for (int i = 0; i < 4; ++i) HGR_STREAM_STORE((f32x4 *)(row + i * 16), acc[i][j]);
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
                    if (n + 3 < nend) HGR_STREAM_STORE((f32x4 *)(row + n), acc[i][j]);
                    else
                        for (int e = 0; e < 4; ++e) if (n + e < nend) row[n + e] = acc[i][j][e];
                }
            }
        }
    }
}


// =================================================================================================================
// gemm_nt_wres: the same product with the roles turned around - W RESIDENT, A streamed.
//
// What the ring kernel above showed (tools/stream_bench.py, tools/ldsdma_bw.hip): its time is ~0.9 us per K-tile whatever the
// depth of the ring, 9.5 us even at M = 64-128 where W crosses the chip once, and the stores add their full HBM time on top
// (15.8 us without them, 22.5 with) - the class embeddings arrive as a chain of dependent, barrier-coupled stages, each a
// round trip to HBM / the MALL.  Here every CU asks for ALL of its class embeddings in one go:
//   * the grid is one workgroup per CU; workgroup c owns 5 or 6 groups of 16 columns (80 / 96 classes), all M <= 512 rows;
//   * its W slice (96 rows x K x 2 B = 96 KB at K = 512) is requested at once by LDS-DMA, K-tile-major and swizzled as in
//     gemm_nt_128: one HBM round trip for the whole kernel, every byte of W crosses the fabric exactly once;
//   * wave w owns rows [64 w, 64 w + 64) x all 96 columns (6 x 4 MFMA tiles, 96 accumulator registers).  Its A rows are
//     shared with nobody in the workgroup, so they never touch the LDS: the lane loads its MFMA fragments straight from
//     global memory (the 512 KB of A are hot in every L2), two K-tiles ahead in a register ring;
//   * per k-step a wave reads 6 W fragments from the LDS and issues 24 v_mfma_f32_16x16x32: 4 x fewer LDS bytes per MFMA
//     than a 64 x 32 wave tile; one barrier in the whole kernel (W has landed).
// K order inside a 64-deep K-tile: the first MFMA takes k = g*16 + 0..7 of every lane group g, the second k = g*16 + 8..15
// (both operands alike), so that the 4 lanes of a row fetch one whole 128-byte line of A with two adjacent 16-byte loads.
// The sum over k is therefore associated differently from gemm_nt_128's: deterministic, equal within fp32 rounding.
constexpr int WR_NT = 6;                   // n tiles (of 16 columns) per workgroup, at most
constexpr int WR_ROWS = WR_NT * 16;        // 96 W rows in the LDS
constexpr int WR_KMAX = 768;               // 96 x 768 x 2 B = 144 KB
constexpr int WR_THREADS = 512;

struct WresArgs {
    const char *A; int64_t lda;
    const char *W; int64_t ldw;
    float *C; int64_t ldc;
    int M, N, K;
    int gq, gr;        // 16-column groups per workgroup: gq + 1 for the first gr workgroups, gq for the others
};

template <int DT>
__global__ __launch_bounds__(WR_THREADS) void gemm_nt_wres(WresArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    __shared__ __attribute__((aligned(1024))) char smem[WR_ROWS * WR_KMAX * 2];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x;
    const int g0 = c * p.gq + min(c, p.gr), ng = p.gq + (c < p.gr ? 1 : 0);
    const int n0 = g0 * 16, nend = min(p.N, n0 + ng * 16);
    if (n0 >= nend) return;
    const int nk = p.K / 64;
    const int KT = WR_ROWS * 128;                  // bytes of one K-tile of the slice

    // W slice -> LDS: piece = 8 rows x 128 B of K-tile kt at smem + kt * KT + (row / 8) * 1024, chunk ^= row & 7
    {
        const int lrow = lane >> 3, ch = (lane & 7) ^ lrow;
        for (int pidx = wave; pidx < nk * (WR_ROWS / 8); pidx += WR_THREADS / 64) {
            const int kt = pidx / (WR_ROWS / 8), pr = pidx - kt * (WR_ROWS / 8);
            const char *src = p.W + (int64_t)min(n0 + pr * 8 + lrow, p.N - 1) * p.ldw * 2 + kt * 128 + ch * 16;
            __builtin_amdgcn_global_load_lds((const AS1 void *)src, (AS3 void *)(smem + kt * KT + pr * 1024), 16, 0, 0);
        }
    }
    const int r = lane & 15, g = lane >> 4;
    const int m0 = wave * 64;
    // lane's A fragment source for m tile j: row (clamped at M - 1, never stored), 32 B at k = g * 16 of every K-tile
    const char *ap[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ap[j] = p.A + (int64_t)min(m0 + j * 16 + r, p.M - 1) * p.lda * 2 + g * 32;
    vec8 af[2][4][2];                              // [K-tile in flight][m tile][k half]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            af[t][j][0] = *(const vec8 *)(ap[j] + t * 128);
            af[t][j][1] = *(const vec8 *)(ap[j] + t * 128 + 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // the W slice is in the LDS; read-only from here on
    if (m0 >= p.M) return;

    const int offW = r * 128;                      // + n tile * 2048 + K-tile * KT
    const int sw0 = ((2 * g) ^ (r & 7)) * 16, sw1 = ((2 * g + 1) ^ (r & 7)) * 16;
    f32x4 acc[WR_NT][4];
#pragma unroll
    for (int i = 0; i < WR_NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const char *wb = smem + (kt0 + t) * KT + offW;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                vec8 wf[WR_NT];
#pragma unroll
                for (int i = 0; i < WR_NT; ++i) wf[i] = *(const vec8 *)(wb + i * 2048 + (h ? sw1 : sw0));
#pragma unroll
                for (int i = 0; i < WR_NT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = T16<DT>::mfma16(wf[i], af[t][j][h], acc[i][j]);
            }
            // this K-tile's slot takes the K-tile two ahead (past the end: a valid address, unused data)
            const int ktn = kt0 + t + 2 < nk ? kt0 + t + 2 : kt0 + t;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                af[t][j][0] = *(const vec8 *)(ap[j] + ktn * 128);
                af[t][j][1] = *(const vec8 *)(ap[j] + ktn * 128 + 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // lane holds, for tile (i, j), C[m0 + j*16 + r][n0 + i*16 + g*4 .. +3]
    if (m0 + 64 <= p.M && n0 + WR_ROWS <= nend) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float *row = p.C + (int64_t)(m0 + j * 16 + r) * p.ldc + n0 + g * 4;
#pragma unroll
            for (int i = 0; i < WR_NT; ++i) HGR_STREAM_STORE((f32x4 *)(row + i * 16), acc[i][j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + j * 16 + r;
            if (m >= p.M) continue;
            float *row = p.C + (int64_t)m * p.ldc;
#pragma unroll
            for (int i = 0; i < WR_NT; ++i) {
                const int n = n0 + i * 16 + g * 4;
                if (n + 3 < nend) HGR_STREAM_STORE((f32x4 *)(row + n), acc[i][j]);
                else
                    for (int e = 0; e < 4; ++e) if (n + e < nend) row[n + e] = acc[i][j][e];
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
    static int cus = 0, kind = -1;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus = n;
    }
    if (kind < 0) { const char *e = getenv("HGR_GEMM_STREAM"); kind = e ? atoi(e) : 1; }
    const int groups16 = (N + 15) / 16;
    if (kind == 2 && M <= 512 && K % 128 == 0 && K <= WR_KMAX && groups16 <= WR_NT * cus) {
        WresArgs w;
        w.A = (const char *)A; w.lda = lda; w.W = (const char *)W; w.ldw = ldw; w.C = C; w.ldc = ldc;
        w.M = M; w.N = N; w.K = K;
        w.gq = groups16 / cus; w.gr = groups16 % cus;
        if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_wres<HGR_BF16>), dim3(cus), dim3(WR_THREADS), 0, (hipStream_t)stream, w);
        else hipLaunchKernelGGL((gemm_nt_wres<HGR_F16>), dim3(cus), dim3(WR_THREADS), 0, (hipStream_t)stream, w);
        return 1;
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
