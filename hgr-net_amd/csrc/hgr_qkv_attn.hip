// =================================================================================================
// hgr_gemm_nt_ln_mha: the attention half of a residual block in ONE launch -
//     att = softmax(q k^T / 8 [+ causal mask]) v,   [q | k | v] = LayerNorm(x) W_in^T + b_in
// (clip/model.py:167-188: ln_1 -> nn.MultiheadAttention's in_proj -> scaled dot-product attention; out_proj stays a GEMM of its own).
// The [M, 3 W] qkv tensor is never written or re-read: at ViT-B/32 batch 512 that is 118 MB out + 118 MB in per layer, and the
// separate attention launch (hgr_mha: 31.7 us of a 134 us pair) disappears.
//
// Tile = 256 rows x 192 columns x K: the rows are S = floor(256 / L) WHOLE sequences (L <= 64 tokens: 5 x 50 at ViT-B/32), the
// columns the 64 q, 64 k and 64 v outputs of ONE head, so softmax(q k^T) v of those S sequences and that head needs nothing from
// outside the tile.  512 threads = 8 waves as 4 (M) x 2 (N); a wave owns 64 x 96 = 4 x 6 MFMA tiles = 96 accumulator registers.
// One workgroup per CU (LDS 112 KB; two waves per SIMD).
//
// Main loop = the LayerNorm-folded consumer GEMM of gemm_nt_duo (same operand roles, same K order per output element, hence the same
// accumulator bits).  A K-tile (64 deep) is staged by LDS-DMA as three pieces cut by the phase that reads them:
//     PA0 = rows 0-31 of every wave row (16 KB), PW = the 192 weight rows (24 KB): read in ph1;   PA1 = rows 32-63 (16 KB): read in ph2
// two stages of 56 KB; a piece is refilled for K-tile t + 2 as soon as both ping-pong groups have read it:
//     ph1(t): issues PA1(t+1) x2                reads W (12 x ds_read_b128), A rows 0-31 (4 x)   waits vmcnt(7): PA1(t) landed
//     ph2(t): issues PA0(t+2) x2, PW(t+2) x3    reads A rows 32-63 (4 x)                          waits vmcnt(7): PA0(t+1), PW(t+1) landed
// (xN = global_load_lds_dwordx4 per thread; a count = "my N youngest may still be in flight"), 24 MFMAs per wave and phase.  Waves
// 0-3 and 4-7 (one of each per SIMD) run one barrier interval apart, so one group's reads and waits sit beside the other's MFMAs
// (the ping-pong of gemm_nt_256).
//
// Epilogue: y = rstd (acc - mean s_n) + c_n as in the consumer GEMM, rounded to the MFMA type - the value the unfused path stores in
// qkv - into LDS: Q and K as swizzled 128-byte rows, V row-major (144-byte rows).  Then hgr_mha's arithmetic, instruction for
// instruction, per 16-query tile and sequence: S^T = K Q^T on the matrix cores (keys from LDS as the A operand, queries as B),
// softmax in registers, the exponentiated accumulators as the B operand of O^T = V^T P^T (V^T fragments by ds_read_b64_tr_b16).  A
// query tile that straddles two sequences is computed against each and every lane keeps its own sequence's result.  Same bits as
// hgr_gemm_nt_ln + hgr_mha (tests/test_gpu_kernels.py::test_gemm_ln_mha_equals_gemm_then_mha).
// =================================================================================================
#include "hgr_gemm_common.h"

namespace hgr_gemm {

struct QkvAttnArgs {
    const char *A; int64_t lda;          // x16 [M, K]: the un-normalised 16-bit rows (residual stream's high half)
    const char *W; int64_t ldw;          // gamma-folded in_proj weight [3 Wd, K]
    const float *ln_s, *ln_c;            // [3 Wd]
    const float *ln_stats; int ln_slots; float ln_eps;
    void *out; int64_t ldo;              // att [M, Wd] 16-bit
    int M, K, Wd, L, H, S;               // S = sequences per tile, rows per tile = S * L
    int tiles_m;
    int hsplit;                          // raster: 1 = an XCD walks whole row tiles (all heads); 2 = the XCDs in two head halves x four row quarters
};

namespace {
typedef __attribute__((ext_vector_type(4))) short qa_s16x4;
typedef __attribute__((ext_vector_type(8))) short qa_s16x8;

// a phase's LDS-DMAs are issued BEHIND its fragment reads (as in gemm_nt_duo): an LDS-DMA instruction can stall at the CU's one
// vector-memory address path, and in front of the reads that stall delayed the reads, the wait and the MFMA burst behind them.  Three
// interleaved pairs, one box: 110.1 -> 107.9 us per launch, bit-identical.  -DHGR_QA_ISSUE_LATE=0: the round-4 order
#ifndef HGR_QA_VASM
#define HGR_QA_VASM 1
#endif
// Experiment, OFF (-DHGR_QA_STATS_DMA=1 builds it): the tile's raw slot statistics staged by LDS-DMA into the V region (unused while the
// main loop runs) during the second-last K-tile - the six DMAs per wave (waves 0-3) ride in the in-order vmcnt stream of the main loop,
// the two counted waits of that K-tile allow for them, the last K-tile's vmcnt(0) completes them - instead of a dependent global round
// trip between the main loop and the q / k / v conversion.  Measured twice in round 6 (tools/qa_stamps.py + three interleaved rounds of
// the image tower each): (a) issued behind the seam barrier: statistics section 1 880 -> 1 160 ticks of a 37 400-tick tile, launch
// 107.6 -> 108.6 us, tower 4.78 -> 4.81 ms; (b) issued mid-loop as built here: section 1 880 -> 1 160 again, main loop 23 050 -> 24 000,
// tile 37 364 -> 37 664, tower 4.78 -> 4.80 ms.  The round trip is real (5 % of a tile) but whatever removes it gives the time back
// elsewhere; round 4 found the same with a third placement.  Under the package power limit idle cycles are not free
// time to be harvested: the clock drops when they are filled.
#ifndef HGR_QA_STATS_DMA
#define HGR_QA_STATS_DMA 0
#endif
#ifndef HGR_QA_ISSUE_LATE
#define HGR_QA_ISSUE_LATE 1
#endif
constexpr int QA_NT = 512;
constexpr int QA_STAGE = 57344;                              // PA0 16 K | PA1 16 K | PW 24 K
constexpr int QA_PA0 = 0, QA_PA1 = 16384, QA_PW = 32768;
// attention-phase layout (104 448 B) ABOVE stage 0: a persistent workgroup lets the next tile's first K-tile land in stage 0 meanwhile
constexpr int QA_VR = 72;
constexpr int QA_Q = QA_STAGE, QA_K = QA_Q + 32768, QA_V = QA_K + 32768, QA_LN = QA_V + 256 * QA_VR * 2;
constexpr int QA_LDS = QA_LN + 2048;                         // 161 792 B of the CU's 163 840
constexpr int QA_ST = QA_V;                                  // staged slot statistics [q < 8][256 rows] x 16 B = 32 KB of V's 36 KB (HGR_QA_STATS_DMA)

#ifdef HGR_LAB
// lab builds only: s_memtime stamps of wave 0 around the sections of the workgroup's SECOND tile (steady state), read back through
// hgr_lab_qa_stamps (tools/qa_stamps.py): 0 tile start, 1 main loop done, 2 row statistics in LDS, 3 q / k / v in LDS, 4 attention done
__device__ unsigned long long qa_lab_stamps[1024 * 8];
#define QA_STAMP(i) do { if (threadIdx.x == 0 && lab_tile == 1) { qa_lab_stamps[(blockIdx.x & 1023) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
        if ((i) == 0) qa_lab_stamps[(blockIdx.x & 1023) * 8 + 6] = __builtin_amdgcn_s_memrealtime(); \
        if ((i) == 5) qa_lab_stamps[(blockIdx.x & 1023) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)      /* 6, 7: the 100 MHz clock at stamps 0 and 5 */
#else
#define QA_STAMP(i) do { } while (0)
#endif

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(QA_NT) void qkv_attn(QkvAttnArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    __shared__ __attribute__((aligned(1024))) char smem[QA_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;          // waves w and w + 4 share a SIMD: wn is also the ping-pong group
    const int r = lane & 15, g = lane >> 4;

    // tiles -> workgroups.  Tile id = row tile * H + head (heads fastest).  Every XCD (blocks b, b + 8, ... share an L2) owns a
    // contiguous range of tile ids and its workgroups walk it with a stride of the XCD's workgroup count: the tiles in flight on an L2
    // are a contiguous window - the 12 heads of ~3 row panels beside the 3.5 MB of folded weights.  grid = tiles: one tile per
    // workgroup.  grid = CUs (the default): PERSISTENT workgroups; the next tile's first K-tile is requested before this tile's
    // attention phase, so its prologue latency (and a workgroup launch) is off the critical path.
    // p.hsplit == 2: XCD x owns the head half x & 1 of the row tiles of quarter x >> 1 - its L2 then holds HALF of the folded weights
    // (1.8 MB at 12 heads) beside the row panels in flight, instead of re-reading all of them in every round of tiles.
    const int ntiles = p.tiles_m * p.H, orig = blockIdx.x, G = gridDim.x;
    const int xcd = orig & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
    const bool split = p.hsplit == 2;
    const int mlo = split ? (xcd >> 1) * p.tiles_m / 4 : 0;
    const int HS = split ? p.H >> 1 : p.H, hbase = split ? (xcd & 1) * HS : 0;
    const int xbase = split ? 0 : xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcnt = split ? (((xcd >> 1) + 1) * p.tiles_m / 4 - mlo) * HS : q8 + (xcd < r8 ? 1 : 0);
    const int xwgs = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    int cur = orig >> 3;                                     // index of this workgroup's tile inside the XCD's range
    if (cur >= xcnt) return;
    const int rows_valid = p.S * p.L;

    // per-lane source offsets of one K-tile's LDS-DMA instructions (bytes from A / W; operands < 4 GB, checked on the host)
    unsigned oA0[2], oA1[2], oW[3];
    int m0, h;
    auto set_tile = [&](int tile) {
        const int tq = tile / HS;
        h = hbase + tile - tq * HS;
        m0 = (mlo + tq) * rows_valid;
        // opaque copy: every lane-derived term below is recomputed per tile instead of being kept live (and spilled) across the main loop
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int id = (i * 8 + wave) * 64 + ln;
            const int pr = id >> 3, c = (id & 7) ^ (pr & 7);
            if (i < 2) {
                const int tr = (pr >> 5) * 64 + (pr & 31);
                oA0[i] = (unsigned)(((int64_t)min(m0 + tr, p.M - 1) * p.lda + c * 8) * 2);
                oA1[i] = (unsigned)(((int64_t)min(m0 + tr + 32, p.M - 1) * p.lda + c * 8) * 2);
            }
            const int wr = (pr >> 6) * p.Wd + h * 64 + (pr & 63);      // piece row pr = column pr of the tile: q 0-63, k 64-127, v 128-191
            oW[i] = (unsigned)(((int64_t)wr * p.ldw + c * 8) * 2);
        }
    };
    set_tile(xbase + cur);
    char *const ldsw = smem + wave * 1024;
    const __amdgpu_buffer_rsrc_t rA = dma_rsrc(p.A), rW = dma_rsrc(p.W);
    auto issueA = [&](const unsigned (&off)[2], int piece, int t) {
        char *dst = ldsw + (t & 1) * QA_STAGE + piece;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rA, p.A, off[i], t * 128, dst + i * 8192);
    };
    auto issueW = [&](int t) {
        char *dst = ldsw + (t & 1) * QA_STAGE + QA_PW;
#pragma unroll
        for (int i = 0; i < 3; ++i) dma16(rW, p.W, oW[i], t * 128, dst + i * 8192);
    };
    // Slot statistics of the tile at m0 (waves 0-3: row 64 wave + lane, entry q = slots 2 q, 2 q + 1) -> LDS [q][row], issued at the top
    // of the second-last K-tile; read back by the wave that issued it (thread t: row t).
    const int nq = p.ln_slots >> 1;
    const bool st_dma = HGR_QA_STATS_DMA && !(p.ln_slots & 1) && (nq == 2 || nq == 4 || nq == 5 || nq == 6 || nq == 8);
    const __amdgpu_buffer_rsrc_t rS = dma_rsrc(p.ln_stats);
    auto issueStats = [&]() {
        if (!st_dma || wave >= 4) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const unsigned so = (unsigned)min(m0 + wave * 64 + ln, p.M - 1) * (unsigned)(p.ln_slots * 8);
        char *dst = smem + QA_ST + wave * 1024;
        for (int q = 0; q < nq; ++q) dma16(rS, (const char *)p.ln_stats, so, q * 16, dst + q * 4096);
    };

    const int nk = p.K / 64;    // >= 2 (host guarantees)
    const int offA = (wm * 32 + r) * 128;          // + m tile (0, 1) * 2048 within PA0 / PA1
    const int offW = (wn * 96 + r) * 128;          // + n tile * 2048 within PW
    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;
    vec8 wf[6][2], af[2][2];

    // prologue in steady-state order: PA0(0), PW(0) | PA1(0) | PA0(1), PW(1)
    issueA(oA0, QA_PA0, 0); issueW(0);
    issueA(oA1, QA_PA1, 0);
    issueA(oA0, QA_PA0, 1); issueW(1);
#ifdef HGR_LAB
  int lab_tile = -1;
#endif
  for (;;) {
#ifdef HGR_LAB
    ++lab_tile;
#endif
    QA_STAMP(0);
    f32x4 acc[4][6];            // [m tile][n tile]: C[wm*64 + 16 i + r][wn*96 + 16 j + 4 g .. + 3]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // PA0(0), PW(0) landed: my 7 youngest operations are PA1(0) x2 + PA0(1), PW(1) x5 (first tile), or - across a tile seam, where the
    // previous tile's output stores sit between PA1(0) and PA0(1) in issue order - PA0(1), PW(1) x5 and two of those stores
    HGR_RWAIT(7);
    if (wn) HGR_MBAR();         // ping-pong: group 1 runs one barrier interval behind group 0

    // MODE 0: steady state (t + 2 < nk), 1: second-last K-tile, 2: last K-tile
    auto ktile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char *st = smem + (t & 1) * QA_STAGE;
        if (MODE == 1) issueStats();          // m0 is still this tile's (set_tile for the next one runs in the epilogue); V is free: the seam barrier is behind us
        // ---- ph1: rows 0-31 of the wave x all 96 columns ----
#if !HGR_QA_ISSUE_LATE
        if (MODE <= 1) issueA(oA1, QA_PA1, t + 1);              // its slot was last read in ph2(t - 1), two barriers ago
#endif
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            wf[j][0] = *(const vec8 *)(st + QA_PW + offW + j * 2048 + sw0);
            wf[j][1] = *(const vec8 *)(st + QA_PW + offW + j * 2048 + sw1);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i][0] = *(const vec8 *)(st + QA_PA0 + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(st + QA_PA0 + offA + i * 2048 + sw1);
        }
#if HGR_QA_ISSUE_LATE
        if (MODE <= 1) issueA(oA1, QA_PA1, t + 1);
#endif
        if (MODE == 1 && st_dma && wave < 4) {                  // + this wave's nq statistics DMAs, issued at the top of this K-tile
            switch (nq) { case 2: HGR_RWAIT(9); break; case 4: HGR_RWAIT(11); break; case 5: HGR_RWAIT(12); break; case 6: HGR_RWAIT(13); break; default: HGR_RWAIT(15); }
        } else if (MODE <= 1) HGR_RWAIT(7); else HGR_RWAIT(0);  // PA1(t) landed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[i][j] = T16<DT>::mfma16(wf[j][kk], af[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph2: rows 32-63 ----
#if !HGR_QA_ISSUE_LATE
        if (MODE == 0) { issueA(oA0, QA_PA0, t + 2); issueW(t + 2); }     // read in ph1(t), two barriers ago
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i][0] = *(const vec8 *)(st + QA_PA1 + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(st + QA_PA1 + offA + i * 2048 + sw1);
        }
#if HGR_QA_ISSUE_LATE
        if (MODE == 0) { issueA(oA0, QA_PA0, t + 2); issueW(t + 2); }
#endif
        if (MODE == 1 && st_dma && wave < 4) {
            switch (nq) { case 2: HGR_RWAIT(4); break; case 4: HGR_RWAIT(6); break; case 5: HGR_RWAIT(7); break; case 6: HGR_RWAIT(8); break; default: HGR_RWAIT(10); }
        } else if (MODE == 0) HGR_RWAIT(7); else if (MODE == 1) HGR_RWAIT(2); else HGR_RBAR();      // PA0(t+1), PW(t+1) landed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[2 + i][j] = T16<DT>::mfma16(wf[j][kk], af[i][kk], acc[2 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
    };
    for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>());
    ktile(nk - 2, std::integral_constant<int, 1>());
    ktile(nk - 1, std::integral_constant<int, 2>());
    if (!wn) HGR_MBAR();        // group 0 waits for group 1's last interval: every LDS read is done, no DMA in flight
    QA_STAMP(1);
    // this tile's coordinates for the epilogue; then (persistent) the next tile's first K-tile is requested into stage 0, which the
    // attention phase below does not touch
    const int m0c = m0, hc = h;
    const int nxt = cur + xwgs;
    const bool has_next = nxt < xcnt;

    // the folded-LayerNorm vectors of this lane's columns, two n tiles ahead of their use (round 5: all six pairs at once were 48
    // registers on top of the 96 accumulators - the kernel spilled, and every reload waited with vmcnt(0) behind whatever stores were
    // in flight); the first two pairs are requested here, beside the statistics loads and ahead of the next tile's DMAs
    auto ln_vec = [&](int j, f32x4 &sv, f32x4 &cv) {
        const int col0 = wn * 96 + j * 16;                           // wave-uniform: part 0 = q, 1 = k, 2 = v; head dim d0 + 4 g + e
        const int gn = (col0 >> 6) * p.Wd + hc * 64 + (col0 & 63) + g * 4;
        sv = *(const f32x4 *)(p.ln_s + gn);
        cv = *(const f32x4 *)(p.ln_c + gn);
    };
    f32x4 sq[3], cq[3];                          // ring of three: n tile j in slot j % 3
    ln_vec(0, sq[0], cq[0]);
    ln_vec(1, sq[1], cq[1]);
    // ---- LayerNorm row statistics of the tile's 256 rows (thread t and t + 256: row t), as the consumer GEMM finalises them ----
    // They are loaded, summed and written BEFORE the next tile's DMAs are issued (round 5): vmcnt retires in issue order, so behind
    // the DMAs the wait for the statistics was a wait for the next tile's whole first K-tile.  And every LDS access from here to the
    // barrier in front of the attention phase is inline asm: hipcc orders an LDS access it can see behind ALL LDS-DMAs in flight
    // (vmcnt(0)) - the DMAs now land under the q / k / v conversion instead of being waited for in front of it.
    typedef __attribute__((ext_vector_type(2))) float qa_f2;
    const unsigned lnrow_a = (unsigned)(uintptr_t)(AS3 char *)(smem + QA_LN);
    int tid_e = threadIdx.x;                     // (opaque: the row index of the statistics is recomputed here, not carried through the main loop)
    asm volatile("" : "+v"(tid_e));
    if (tid_e < 256) {
        const f32x4 *sp = (const f32x4 *)(p.ln_stats + (int64_t)min(m0c + tid_e, p.M - 1) * p.ln_slots * 2);
        float s1 = 0.f, s2 = 0.f;
        if (st_dma) {
            // four entries at a time, summed in slot order like the loads of the other path (entries >= nq are whatever V held: read,
            // never summed); inline asm: an LDS read the compiler can see is ordered behind every LDS-DMA in flight
            const unsigned sa = (unsigned)(uintptr_t)(AS3 char *)(smem + QA_ST) + (unsigned)tid_e * 16u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 tl[4];
                if (h == 0) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288\n\ts_waitcnt lgkmcnt(0)"
                                         : "=&v"(tl[0]), "=&v"(tl[1]), "=&v"(tl[2]), "=&v"(tl[3]) : "v"(sa) : "memory");
                else asm volatile("ds_read_b128 %0, %4 offset:16384\n\tds_read_b128 %1, %4 offset:20480\n\tds_read_b128 %2, %4 offset:24576\n\tds_read_b128 %3, %4 offset:28672\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(tl[0]), "=&v"(tl[1]), "=&v"(tl[2]), "=&v"(tl[3]) : "v"(sa) : "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (h * 4 + i < nq) { s1 += tl[i][0] + tl[i][2]; s2 += tl[i][1] + tl[i][3]; }
                if (nq <= 4) break;
            }
        } else {
        auto fixed = [&](auto nq_tag) {
            constexpr int NQ = decltype(nq_tag)::value;
            f32x4 t[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) t[i] = sp[i];
#pragma unroll
            for (int i = 0; i < NQ; ++i) { s1 += t[i][0] + t[i][2]; s2 += t[i][1] + t[i][3]; }
        };
        switch (p.ln_slots) {
            case 4: fixed(std::integral_constant<int, 2>()); break;
            case 8: fixed(std::integral_constant<int, 4>()); break;
            case 10: fixed(std::integral_constant<int, 5>()); break;
            case 12: fixed(std::integral_constant<int, 6>()); break;
            case 16: fixed(std::integral_constant<int, 8>()); break;
            default:
                for (int i = 0; i < p.ln_slots / 2; ++i) { const f32x4 t = sp[i]; s1 += t[0] + t[2]; s2 += t[1] + t[3]; }
        }
        }
        const float2 mr = ln_finalize(s1, s2, 1.0f / (float)p.K, p.ln_eps);
        const qa_f2 mv = {mr.x, mr.y};
        asm volatile("ds_write_b64 %0, %1" ::"v"(lnrow_a + tid_e * 8), "v"(mv) : "memory");
    }
    HGR_RBAR();
    QA_STAMP(2);

    // ---- q / k / v of the tile, rounded to the MFMA type, into LDS ----
    qa_f2 mrv[4];
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:128\n\tds_read_b64 %2, %4 offset:256\n\tds_read_b64 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(mrv[0]), "=&v"(mrv[1]), "=&v"(mrv[2]), "=&v"(mrv[3]) : "v"(lnrow_a + (wm * 64 + r) * 8) : "memory");
    const unsigned smem_a = (unsigned)(uintptr_t)(AS3 char *)smem;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        if (j + 2 < 6) ln_vec(j + 2, sq[(j + 2) % 3], cq[(j + 2) % 3]);
        const int col0 = wn * 96 + j * 16;
        const int part = col0 >> 6, d = (col0 & 63) + g * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 64 + i * 16 + r;
            const float2 mri = make_float2(mrv[i][0], mrv[i][1]);
            // branch-free destination: V rows are 144 bytes, row-major; Q / K rows 128 bytes with the chunk swizzle of the fragment reads
            const int base = part == 2 ? QA_V : part ? QA_K : QA_Q;
            const int at = part == 2 ? row * (QA_VR * 2) + d * 2 : row * 128 + (((d >> 3) ^ (row & 7)) * 16) + (d & 4) * 2;
            const u32x2 pk = ln_out16<DT, false>(mri, acc[i][j], sq[j % 3], cq[j % 3]);        // the consumer GEMM's rounding (hgr_gemm_common.h)
            asm volatile("ds_write_b64 %0, %1" ::"v"(smem_a + base + at), "v"(pk) : "memory");
        }
    }
    if (has_next) {
        // (persistent) the next tile's first K-tile is requested into stage 0, which the attention phase below does not touch - HERE,
        // behind the last wait for a global load of the epilogue: hipcc's counts for ordinary loads leave the LDS-DMAs out, so a wait
        // for ANY load issued around the DMAs is a wait for the DMAs (vmcnt retires in issue order)
        __builtin_amdgcn_sched_barrier(0);
        set_tile(xbase + nxt);
        issueA(oA0, QA_PA0, 0); issueW(0);
        issueA(oA1, QA_PA1, 0);
    }
    HGR_RBAR();
    QA_STAMP(3);

    // ---- attention: query tile qt = rows 16 qt .. 16 qt + 15 of the tile; wave w takes tiles w and w + 8 ----
    // (round 5, measured and not kept: the (query tile, sequence) units dealt round-robin over the waves and worked two or three at a
    // time with interleaved instruction streams - 114 us -> 115 / 118 us per launch, one at a time 120: the phase is not bound by the
    // latency of a unit's dependent chain; profiles/NOTES.md)
    // (also measured and not kept: hipcc holds the seven next-tile DMAs above as pending through this loop and puts s_waitcnt vmcnt(0) in
    // front of the transposing V reads of every pass - a wait for the previous query tile's output stores too.  With those DMAs
    // written as inline asm the wait is gone and the launch takes 109.6 us against 108.7: it was never exposed)
    constexpr int KT = 2;              // 64 key slots: L <= 64
    // Round 6 (tools/qa_stamps.py: the phases behind the main loop were 39 % of a tile - statistics 5, q / k / v 11, attention 14 + 9 at
    // the seam barrier for the waves that have three units): a unit = (16-query tile, sequence) is a chain of LDS and cross-lane
    // latencies, not of arithmetic - ~2 800 cycles for 16 MFMAs.  Same arithmetic, element for element (hgr_mha's), laid out for latency:
    //   * ALL of a unit's fragment reads - 8 K reads and the 16 transposing V reads - are issued before its first MFMA (the compiler's
    //     own order was read 2 / wait / 2 MFMAs, four times, then the V reads behind the softmax);
    //   * the row maximum and the row sum cross the four lane groups through v_permlane16_swap / v_permlane32_swap (vector ALU)
    //     instead of four dependent ds_bpermute round trips; a + b of the SAME pair in both lanes is the same sum in either order;
    //   * the lane-constant part of the key mask (slots past the sequence) is an additive 0 / -inf vector (one FMA with the exact
    //     scaling by 1/8 where multiply, compare and select were three instructions).
    f32x4 kbias[2 * KT];
    int ln_a = lane;                     // opaque copy: lane constants the compiler would otherwise hoist out of the tile loop and carry (spill) through the main loop
    asm volatile("" : "+v"(ln_a));
    const int g_a = ln_a >> 4;
#pragma unroll
    for (int t = 0; t < 2 * KT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) kbias[t][e] = (t * 16 + g_a * 4 + e >= p.L) ? -INFINITY : 0.f;
    auto x16max = [](float v) { auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])); };
    auto x32max = [](float v) { auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])); };
    auto x16add = [](float v) { auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); return __uint_as_float(a[0]) + __uint_as_float(a[1]); };
    auto x32add = [](float v) { auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); return __uint_as_float(a[0]) + __uint_as_float(a[1]); };
    AS3 const char *const ldsV = (AS3 const char *)(smem + QA_V);
    for (int qt = wave; qt * 16 < rows_valid; qt += 8) {
        const int qrow = qt * 16 + r;                                // this lane's query row inside the tile
        const int myseq = min(qrow / p.L, p.S - 1);
        const int s_lo = (qt * 16) / p.L, s_hi = min((qt * 16 + 15) / p.L, p.S - 1);
        const char *qr = smem + QA_Q + qrow * 128;
        const vec8 q0 = *(const vec8 *)(qr + sw0), q1 = *(const vec8 *)(qr + sw1);
        f32x4 res[4];
        float rinv = 0.f;
#pragma unroll
        for (int td = 0; td < 4; ++td) res[td] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = s_lo; s <= s_hi; ++s) {
            const int kb = s * p.L;                                  // first key row of sequence s
            const int q = qrow - kb;                                 // query position inside the sequence (meaningful when myseq == s)
            // ---- every fragment of the unit: K (A operand of S^T = K Q^T), V^T (A operand of O^T = V^T P^T; hgr_mha's transposing reads:
            //      V^T[d = 16 td + r][keys 32 ks + 4 g .. + 3, 32 ks + 16 + 4 g .. + 3]) ----
            vec8 kf[2 * KT][2];
#pragma unroll
            for (int t = 0; t < 2 * KT; ++t) {
                const int krow = min(kb + t * 16 + r, 255);          // key slots past the sequence are masked below: any finite row will do
                const char *kr = smem + QA_K + krow * 128;
                kf[t][0] = *(const vec8 *)(kr + (((0 + g) ^ (krow & 7)) * 16));
                kf[t][1] = *(const vec8 *)(kr + (((4 + g) ^ (krow & 7)) * 16));
            }
            qa_s16x4 vlo[KT][4], vhi[KT][4];
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                const int v0 = min(kb + ks * 32 + g * 4 + (r >> 2), 255), v1 = min(kb + ks * 32 + 16 + g * 4 + (r >> 2), 255);
#if HGR_QA_VASM
                // as inline asm: hipcc puts s_waitcnt vmcnt(0) in front of a transposing read it can see while LDS-DMAs are pending (the
                // next tile's first K-tile, requested above, lands in stage 0 - never in V) - here that is also a wait for the previous
                // unit's output stores.  The reads are waited for by hand in front of the P V products (QA_V_WAIT).
                const unsigned a0 = (unsigned)(uintptr_t)ldsV + (unsigned)((v0 * QA_VR + (r & 3) * 4) * 2), a1 = (unsigned)(uintptr_t)ldsV + (unsigned)((v1 * QA_VR + (r & 3) * 4) * 2);
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:32\n\tds_read_b64_tr_b16 %2, %4 offset:64\n\tds_read_b64_tr_b16 %3, %4 offset:96"
                             : "=&v"(vlo[ks][0]), "=&v"(vlo[ks][1]), "=&v"(vlo[ks][2]), "=&v"(vlo[ks][3]) : "v"(a0));
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:32\n\tds_read_b64_tr_b16 %2, %4 offset:64\n\tds_read_b64_tr_b16 %3, %4 offset:96"
                             : "=&v"(vhi[ks][0]), "=&v"(vhi[ks][1]), "=&v"(vhi[ks][2]), "=&v"(vhi[ks][3]) : "v"(a1));
#else
#pragma unroll
                for (int td = 0; td < 4; ++td) {
                    vlo[ks][td] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 qa_s16x4 *)(ldsV + (v0 * QA_VR + td * 16 + (r & 3) * 4) * 2));
                    vhi[ks][td] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 qa_s16x4 *)(ldsV + (v1 * QA_VR + td * 16 + (r & 3) * 4) * 2));
                }
#endif
            }
            f32x4 sc[2 * KT];
#pragma unroll
            for (int t = 0; t < 2 * KT; ++t) sc[t] = T16<DT>::mfma16(kf[t][0], q0, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int t = 0; t < 2 * KT; ++t) sc[t] = T16<DT>::mfma16(kf[t][1], q1, sc[t]);
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2 * KT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = __builtin_fmaf(sc[t][e], 0.125f, kbias[t][e]);      // 64^-0.5, exact; -inf on the slots past the sequence
                    if (CAUSAL && t * 16 + g_a * 4 + e > q) x = -INFINITY;
                    sc[t][e] = x;
                    mx = fmaxf(mx, x);
                }
            mx = x32max(x16max(mx));
            float sum = 0.f;
            const float nmx = -mx * 1.4426950408889634f;                          // e^(s - mx) = 2^(s log2e - mx log2e): FMA + v_exp_f32, as hgr_mha
#pragma unroll
            for (int t = 0; t < 2 * KT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pexp = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][e], 1.4426950408889634f, nmx));
                    sc[t][e] = pexp;
                    sum += pexp;
                }
            sum = x16add(sum);
            sum = x32add(sum);
            const float inv = 1.0f / sum;
            f32x4 o[4];
#pragma unroll
            for (int td = 0; td < 4; ++td) o[td] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if HGR_QA_VASM
            // every V^T fragment has landed (they were issued before the softmax); the registers pass through the statement so that no
            // product can be scheduled in front of it
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0][0]), "+v"(vlo[0][1]), "+v"(vlo[0][2]), "+v"(vlo[0][3]), "+v"(vhi[0][0]), "+v"(vhi[0][1]), "+v"(vhi[0][2]), "+v"(vhi[0][3]),
                                                  "+v"(vlo[1][0]), "+v"(vlo[1][1]), "+v"(vlo[1][2]), "+v"(vlo[1][3]), "+v"(vhi[1][0]), "+v"(vhi[1][1]), "+v"(vhi[1][2]), "+v"(vhi[1][3]));
#endif
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                vec8 pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pf[e] = (E)sc[2 * ks][e]; pf[4 + e] = (E)sc[2 * ks + 1][e]; }
#pragma unroll
                for (int td = 0; td < 4; ++td) {
                    const vec8 vf = __builtin_bit_cast(vec8, (qa_s16x8)__builtin_shufflevector(vlo[ks][td], vhi[ks][td], 0, 1, 2, 3, 4, 5, 6, 7));
                    o[td] = T16<DT>::mfma16(vf, pf, o[td]);
                }
            }
            if (s_lo == s_hi) {                 // (wave-uniform) the query tile lies inside one sequence: no per-lane selection, no copies
                rinv = inv;
#pragma unroll
                for (int td = 0; td < 4; ++td) res[td] = o[td];
            } else if (myseq == s) {
                rinv = inv;
#pragma unroll
                for (int td = 0; td < 4; ++td) res[td] = o[td];
            }
        }
        if (qrow < rows_valid && m0c + qrow < p.M) {
            E *orow = (E *)p.out + (int64_t)(m0c + qrow) * p.ldo + hc * 64 + g * 4;
#pragma unroll
            // product and 16-bit conversion as ONE operation, as in hgr_mha (mul_pack16: v_fma_mixlo / mixhi_f16 - one rounding)
            for (int td = 0; td < 4; ++td)
                *(u32x2 *)(orow + td * 16) = (u32x2){mul_pack16<DT>(res[td][0], rinv, res[td][1], rinv), mul_pack16<DT>(res[td][2], rinv, res[td][3], rinv)};
        }
    }
    QA_STAMP(4);
    if (!has_next) break;
    __syncthreads();            // every wave's attention reads are done: stage 1 (under Q / K) may be refilled, V is free
    QA_STAMP(5);
    issueA(oA0, QA_PA0, 1); issueW(1);
    cur = nxt;
  }
}
}  // namespace

// HGR_QA_PERSIST=0: one tile per workgroup (grid = tiles); default: one persistent workgroup per CU
static int qa_persist() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("HGR_QA_PERSIST"); v = e ? atoi(e) : 1; }
    return v;
}
// The two-head-half raster (QkvAttnArgs::hsplit), default on; HGR_QA_HSPLIT=0 = every XCD walks whole row tiles.  Measured on the
// ViT-B/32 evaluation step (two interleaved rounds, one box): counter reads per launch 194 -> 155 MB, 112.3 -> 110.7 us, step 4.951 ->
// 4.928 ms; bit-identical (a tile's arithmetic does not depend on where it runs).  The same split of gemm_nt_duo's raster (two column
// halves x four row quarters for c_fc) changed neither its reads (207 -> 203 MB) nor its time and is not in the tree; profiles/NOTES.md
static int qa_hsplit() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("HGR_QA_HSPLIT"); v = e ? atoi(e) : 1; }
    return v;
}
static int qa_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
        else n = 256;
    }
    return n;
}

void launch_qkv_attn(const QkvAttnArgs &a, int dtype, bool causal, hipStream_t s) {
    const int tiles = a.tiles_m * a.H;
    // (eight-XCD raster; a launch of fewer than 8 row tiles keeps the plain walk)
    const dim3 grid((unsigned)(qa_persist() ? (tiles < qa_cus() ? tiles : qa_cus()) : tiles)), block(QA_NT);
    if (dtype == HGR_BF16) {
        if (causal) hipLaunchKernelGGL((qkv_attn<HGR_BF16, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((qkv_attn<HGR_BF16, false>), grid, block, 0, s, a);
    } else {
        if (causal) hipLaunchKernelGGL((qkv_attn<HGR_F16, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((qkv_attn<HGR_F16, false>), grid, block, 0, s, a);
    }
}

}  // namespace hgr_gemm

using namespace hgr_gemm;

extern "C" int hgr_gemm_nt_ln_mha(const void *X16, int64_t ldx, const void *Wfold, int64_t ldw, const float *ln_s, const float *ln_c,
                                  const float *stats, float eps, void *att, int64_t ldatt, int B, int L, int heads, int causal,
                                  int dtype, void *stream) {
    HGR_REQUIRE(X16 && Wfold && ln_s && ln_c && stats && att, "hgr_gemm_nt_ln_mha: null operand");
    HGR_REQUIRE(B >= 1 && heads >= 1 && L >= 1 && L <= 64, "hgr_gemm_nt_ln_mha: B=%d heads=%d L=%d unsupported (1 <= L <= 64: whole sequences inside a 256-row tile)", B, heads, L);
    const int Wd = heads * 64, K = Wd;
    const int64_t M = (int64_t)B * L;
    HGR_REQUIRE(K % 128 == 0 && K >= 128, "hgr_gemm_nt_ln_mha: the row width %d must be a multiple of 128", K);
    HGR_REQUIRE(ldx >= K && ldw >= K && ldx % 8 == 0 && ldw % 8 == 0 && ldatt >= Wd && ldatt % 4 == 0, "hgr_gemm_nt_ln_mha: leading dimensions (ldx, ldw %% 8 == 0, ldatt %% 4 == 0)");
    HGR_REQUIRE(hgr_aligned(X16, 16) && hgr_aligned(Wfold, 16) && hgr_aligned(ln_s, 16) && hgr_aligned(ln_c, 16) && hgr_aligned(stats, 16) && hgr_aligned(att, 8),
                "hgr_gemm_nt_ln_mha: operands must be 16-byte aligned (att: 8)");
    HGR_REQUIRE(M * ldx * 2 < (1ll << 32) && 3ll * Wd * ldw * 2 < (1ll << 32), "hgr_gemm_nt_ln_mha: operands beyond 4 GB");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_gemm_nt_ln_mha: bad dtype %d", dtype);
    QkvAttnArgs a;
    a.A = (const char *)X16; a.lda = ldx; a.W = (const char *)Wfold; a.ldw = ldw; a.ln_s = ln_s; a.ln_c = ln_c;
    a.ln_stats = stats; a.ln_slots = K / 64; a.ln_eps = eps; a.out = att; a.ldo = ldatt;
    a.M = (int)M; a.K = K; a.Wd = Wd; a.L = L; a.H = heads; a.S = 256 / L; a.tiles_m = (B + a.S - 1) / a.S;
    HGR_REQUIRE((int64_t)a.tiles_m * heads < (1ll << 31), "hgr_gemm_nt_ln_mha: grid too large");
    // ... where the folded weights do not fit an L2 beside the panels in flight anyway (>= 2 MB: width 768 = 3.5 MB); below that the split
    // only reads every row panel on two XCDs (text tower, width 512 = 1.5 MB: 111 -> 188 MB per launch, same time)
    a.hsplit = (qa_hsplit() && heads % 2 == 0 && a.tiles_m >= 8 && 3ll * Wd * K * 2 >= (2ll << 20)) ? 2 : 1;
    launch_qkv_attn(a, dtype, causal != 0, (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_ln_mha");
    return HGR_OK;
}

#ifdef HGR_LAB
extern "C" int hgr_lab_qa_stamps(unsigned long long *host_out, int n_words) {
    if (!host_out || n_words < 1 || n_words > 1024 * 8) return HGR_EINVAL;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hgr_gemm::qa_lab_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? HGR_OK : HGR_ELAUNCH;
}
#endif
