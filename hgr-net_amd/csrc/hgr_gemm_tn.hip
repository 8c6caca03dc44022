// hgr_gemm_tn_splitk: partial[s][a][b] = sum over the rows m of slice s of P[m][a] * Q[m][b]  (fp32 out, 16-bit in).
//
// The weight gradient of a Linear / 1x1 convolution is dW[n][k] = sum_m dY[m][n] X[m][k]: BOTH operands are stored
// with the reduction index m as the slow (row) index.  hgr_gemm_nt wants the reduction index contiguous, which costs a
// transposition of dY and of X per layer and step (and, for 3x3 convolutions, a transposed im2col buffer).  This kernel
// takes the operands as they lie in memory:
//
//   * LDS-DMA (global_load_lds_dwordx4) stages [64 reduction rows][128 columns] tiles, 256-byte rows, both operands;
//   * the MFMA fragment "8 consecutive reduction indices of one output column" is two ds_read_b64_tr_b16 (gfx950's
//     transposing LDS read: a 16-lane group reads a 4-row x 16-column block and lane i receives column i);
//   * the 16-byte chunk XOR `ch ^ (((row & 3) << 2) | ((row >> 2) & 3))` (cdna_hip_programming.md T10, image (b)) is
//     applied on the per-lane SOURCE address of the DMA and on the read address: a 32-lane half reads two blocks 8 rows
//     apart in the same columns, which that image serves without bank conflicts;
//   * rows beyond M and columns beyond Na / Nb are fetched from a zero page, so no operand needs padding;
//   * CONV: Q is the im2col of an NHWC image (3x3, pad 1, stride 1), gathered by the loader - column (tap, c) of pixel
//     row m reads x[m + (ky - 1) W + (kx - 1)][c] or the zero page outside the image: the weight gradient of every 3x3
//     convolution of the ModifiedResNet without an im2col buffer.
//
// Tile 128 (a) x 128 (b) x 64 (m) per stage, 4 waves as 2 x 2, each 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x32, two
// LDS stages (64 KB), blockIdx.y = reduction slice.  EXEC is all ones at every transposing read (no divergent code
// around them), as the ISA requires.
#include "hgr_common.h"
#include <stdlib.h>

#pragma clang diagnostic ignored "-Winline-asm"   // the LDS-DMA assembly below names m0 as clobbered (reserved register: hipcc warns)

namespace {

__device__ __attribute__((aligned(16))) unsigned int tn_zero_page[4] = {0u, 0u, 0u, 0u};

struct TnArgs {
    const char *P; int64_t ldp;      // [M, ldp] 16-bit, Na live columns
    const char *Q; int64_t ldq;      // [M, ldq] 16-bit, Nb live columns (CONV: NHWC image [B, H, W, C], Nb = 9 C)
    float *out; int64_t ldo; int64_t csplit;   // partial s at out + s * csplit, rows a, ldo >= Nb
    int M, Na, Nb, kc;               // slice s covers rows [s * kc, min(M, (s + 1) * kc)), kc % 64 == 0
    int tiles_b;
    // CONV
    int cH, cW, cC;
    unsigned mW, mH, mC;             // ceil(2^32 / W), ceil(2^32 / H), ceil(2^32 / C)
};

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to lds_base + 16 * lane (lds_base wave-uniform).  Issued as
// inline assembly, not __builtin_amdgcn_global_load_lds: behind the builtin hipcc drains s_waitcnt vmcnt(0) in front of the next
// ds_read_b64_tr_b16 of ANY LDS address (it does not for plain ds_read_b128, see gemm_nt_duo), which made "issue the next stage, then
// compute this one" load-then-compute in series in the round-1/2 builds of both kernels of this file.  The waits are this file's own.
__device__ __forceinline__ void glds16(const char *src, char *lds_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(AS3 char *)lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
}

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
// ds_read_b64_tr_b16: the builtin is typed on 16-bit integers; the bits are bf16 / f16 alike
__device__ __forceinline__ s16x4 tr_read(const char *p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4 *)p); }

template <int DT, bool CONV>
__global__ __launch_bounds__(256) void gemm_tn_128(TnArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    constexpr int TILE = 64 * 256;                      // one operand, one stage: 64 rows of 256 B
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int r = lane & 15, g = lane >> 4;
    const int ta = blockIdx.x / p.tiles_b, tb = blockIdx.x - ta * p.tiles_b;
    const int a0 = ta * 128, b0 = tb * 128;
    const int ms = blockIdx.y * p.kc, me = min(p.M, ms + p.kc);
    const int nk = (me - ms + 63) >> 6;

    // LDS-DMA pieces: operand tile = 16 pieces of 1 KB (4 rows); this wave issues pieces i * 4 + wave, i = 0..3
    int prow[4];                     // tile row of this lane's chunk
    int64_t poff[4], qoff[4];        // source byte offset of the chunk inside its row (P) / column offset data (Q)
    bool pok[4], qok[4];
    int qtap_dy[4], qtap_dx[4];      // CONV: tap offsets (ky - 1, kx - 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 4 + (lane >> 4), slot = lane & 15;
        const int ch = slot ^ swz(row);
        prow[i] = row;
        pok[i] = a0 + ch * 8 < p.Na;
        poff[i] = (int64_t)(a0 + ch * 8) * 2;
        const int col = b0 + ch * 8;
        qok[i] = col < p.Nb;
        if (CONV) {
            const int tap = (int)__umulhi((unsigned)col, p.mC), c = col - tap * p.cC;
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
            qtap_dy[i] = ky - 1; qtap_dx[i] = kx - 1;
            qoff[i] = ((int64_t)((ky - 1) * p.cW + (kx - 1)) * p.cC + c) * 2;
        } else {
            qtap_dy[i] = qtap_dx[i] = 0;
            qoff[i] = (int64_t)col * 2;
        }
    }

    auto stage = [&](int buf, int kt) {
        char *sP = smem + buf * (2 * TILE), *sQ = sP + TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = ms + kt * 64 + prow[i];
            const bool in = m < me;
            const char *sp = (in && pok[i]) ? p.P + (int64_t)m * p.ldp * 2 + poff[i] : (const char *)tn_zero_page;
            glds16(sp, sP + (i * 4 + wave) * 1024);
            const char *sq;
            if (CONV) {
                const int t1 = (int)__umulhi((unsigned)m, p.mW), w = m - t1 * p.cW;          // m = (b * H + h) * W + w
                const int t2 = (int)__umulhi((unsigned)t1, p.mH), h = t1 - t2 * p.cH;
                const int hi = h + qtap_dy[i], wi = w + qtap_dx[i];
                const bool ok = in && qok[i] && hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW;
                sq = ok ? p.Q + (int64_t)m * p.cC * 2 + qoff[i] : (const char *)tn_zero_page;
            } else {
                sq = (in && qok[i]) ? p.Q + (int64_t)m * p.ldq * 2 + qoff[i] : (const char *)tn_zero_page;
            }
            glds16(sq, sQ + (i * 4 + wave) * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposing-read addresses: lane l = 4q + pp of group g supplies row r0 + q, chunk c0 + (pp >> 1), half (pp & 1);
    // r0 = kk * 32 + 8 g (+ 4 for the second read), so (row & 3) = q and (row >> 2) & 3 = (2 g + second) & 3.
    const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3;
    int rdoff[2][2];                 // [kk][second read]: byte offset of (row, chunk 0) + the XOR term folded per fragment below
    int rdx[2];                      // XOR term of the row for [second]
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
        rdx[sec] = (q << 2) | ((2 * g + sec) & 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) rdoff[kk][sec] = (kk * 32 + 8 * g + 4 * sec + q) * 256 + 8 * (pp & 1);
    }
    const int chP = wa * 8 + (pp >> 1), chQ = wb * 8 + (pp >> 1);     // + 2 * fragment index

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char *sP = smem + cur * (2 * TILE), *sQ = sP + TILE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            vec8 pf[4], qf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const s16x4 p0 = tr_read(sP + rdoff[kk][0] + 16 * ((chP + 2 * t) ^ rdx[0]));
                const s16x4 p1 = tr_read(sP + rdoff[kk][1] + 16 * ((chP + 2 * t) ^ rdx[1]));
                const s16x4 q0 = tr_read(sQ + rdoff[kk][0] + 16 * ((chQ + 2 * t) ^ rdx[0]));
                const s16x4 q1 = tr_read(sQ + rdoff[kk][1] + 16 * ((chQ + 2 * t) ^ rdx[1]));
                pf[t] = __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7));
                qf[t] = __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T16<DT>::mfma16(qf[i], pf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // lane holds, for tile (i, j): out[a = a0 + wa*64 + j*16 + r][b = b0 + wb*64 + i*16 + 4g .. +3]
    float *o = p.out + (int64_t)blockIdx.y * p.csplit;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int a = a0 + wa * 64 + j * 16 + r;
        if (a >= p.Na) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + wb * 64 + i * 16 + g * 4;
            if (b + 3 < p.Nb) *(f32x4 *)(o + (int64_t)a * p.ldo + b) = acc[i][j];
            else
                for (int e = 0; e < 4; ++e)
                    if (b + e < p.Nb) o[(int64_t)a * p.ldo + b + e] = acc[i][j][e];
        }
    }
}


// gemm_tn_256: the same product on 256 (a) x 256 (b) x 64 (m) stages, 8 waves as 2 (a) x 4 (b), a wave owns 128 x 64 =
// 8 x 4 tiles (128 accumulators).  Why: a 128 x 128 stage holds 2.1 MFLOP of matrix work, ~0.45 us even with two
// workgroups sharing the CU - less than one L2 / Infinity-Cache round trip, so every stage waited for its successor's
// operands.  A 256 x 256 stage is 8.4 MFLOP per 64 KB staged: twice the time for the next stage to land and half the
// L2 -> LDS bytes per FLOP.  The operand tile is stored as two [64][128] halves (all the addressing of gemm_tn_128 carries
// over); 2 stages x 4 halves = 128 KB, one workgroup per CU (2 waves per SIMD).  Linear / 1x1 operands only.
template <int DT>
__global__ __launch_bounds__(512) void gemm_tn_256(TnArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    constexpr int TILE = 64 * 256;                      // one [64][128] half
    __shared__ __attribute__((aligned(1024))) char smem[8 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 2, wb = wave & 3;
    const int r = lane & 15, g = lane >> 4;
    // Workgroup -> (slice, tile), XCD-aware (round 3): the 1-D grid is dealt to the 8 XCDs round-robin (workgroups b, b + 8 share an L2),
    // so XCD x takes the x-th CONTIGUOUS eighth of the (slice, a-tile, b-tile) order: its ~32 workgroups are two or three rows of b-tiles of
    // one slice - 2-3 P panels and one set of Q panels for ~64 panel reads - where dealing tiles b, b + 8, ... of every slice to it made
    // each XCD fetch 7 panels per 12 reads, five slices at once.
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tiles = ((p.Na + 255) >> 8) * p.tiles_b;
    const int slice = wg / tiles, tile = wg - slice * tiles;
    const int ta = tile / p.tiles_b, tb = tile - ta * p.tiles_b;
    const int a0 = ta * 256, b0 = tb * 256;
    const int ms = slice * p.kc, me = min(p.M, ms + p.kc);
    const int nk = (me - ms + 63) >> 6;
    const char *zero_page = (const char *)tn_zero_page;
    asm volatile("" : "+s"(zero_page));                       // keep the address in SGPRs: hipcc re-loaded it from the GOT in every sub-stage

    // Staging (round 3): a RING of four sub-stages of 32 reduction rows (4 x 32 KB = the 128 KB the two 64-row stages used to take), each
    // = 4 operand halves (P0, P1, Q0, Q1) x [32 rows][128 columns] = 8 pieces of 1 KB per half, one per wave: 4 LDS-DMA instructions per
    // thread and sub-stage.  Sub-stage j + 3 is issued while j is computed; waits are COUNTED (vmcnt(8): "all but my 8 youngest DMA
    // instructions have landed" = sub-stage j is in, j + 1 and j + 2 stay in flight) behind one raw s_barrier per sub-stage.  The first
    // version drained vmcnt(0) + __syncthreads() per 64-row stage with ONE stage in flight: every stage waited out most of a memory round
    // trip (26 % of the ViT-L/14 training step sat in this kernel).  Same sums in the same order: bit-identical outputs.
    int64_t poff[4];
    bool pok[4];
    const int lrow = wave * 4 + (lane >> 4);                  // this lane's row inside a sub-stage
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int slot = lane & 15;
        const int ch = slot ^ swz(lrow);
        const int col = (o < 2 ? a0 : b0) + (o & 1) * 128 + ch * 8;
        pok[o] = col < (o < 2 ? p.Na : p.Nb);
        poff[o] = (int64_t)col * 2;
    }
    auto stage = [&](int j) {                                 // sub-stage j -> ring slot j & 3
        char *sb = smem + (j & 3) * (2 * TILE);
        const int m = ms + j * 32 + lrow;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const bool in = m < me && pok[o];
            const char *src = o < 2 ? p.P + (int64_t)m * p.ldp * 2 : p.Q + (int64_t)m * p.ldq * 2;
            const char *sp = in ? src + poff[o] : zero_page;
            glds16(sp, sb + o * (TILE / 2) + wave * 1024);
        }
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3;
    int rdoff[2], rdx[2];
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
        rdx[sec] = (q << 2) | ((2 * g + sec) & 3);
        rdoff[sec] = (8 * g + 4 * sec + q) * 256 + 8 * (pp & 1);
    }
    const int chP = pp >> 1, chQ = (wb & 1) * 8 + (pp >> 1);          // + 2 * fragment index

    const int ns = 2 * nk;                                            // sub-stages (rows beyond the slice come from the zero page)
    stage(0);
    stage(1);
    if (ns > 2) stage(2);
#define HGR_TN_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    for (int j = 0; j < ns; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const int ahead = ns - 1 - j;                                 // sub-stages issued behind j: min(2, ahead)
        if (ahead >= 2) HGR_TN_WAIT(8); else if (ahead == 1) HGR_TN_WAIT(4); else HGR_TN_WAIT(0);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 3 < ns) stage(j + 3);                                 // slot (j - 1) & 3: every wave's reads of it are behind this barrier
        const char *sP = smem + (j & 3) * (2 * TILE) + wa * (TILE / 2), *sQ = smem + (j & 3) * (2 * TILE) + (2 + (wb >> 1)) * (TILE / 2);
        vec8 pf[8], qf[4];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const s16x4 p0 = tr_read(sP + rdoff[0] + 16 * ((chP + 2 * t) ^ rdx[0]));
            const s16x4 p1 = tr_read(sP + rdoff[1] + 16 * ((chP + 2 * t) ^ rdx[1]));
            pf[t] = __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const s16x4 q0 = tr_read(sQ + rdoff[0] + 16 * ((chQ + 2 * t) ^ rdx[0]));
            const s16x4 q1 = tr_read(sQ + rdoff[1] + 16 * ((chQ + 2 * t) ^ rdx[1]));
            qf[t] = __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) acc[i][jj] = T16<DT>::mfma16(qf[i], pf[jj], acc[i][jj]);
        __builtin_amdgcn_s_setprio(0);
    }
#undef HGR_TN_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // lane holds, for tile (i, j): out[a = a0 + wa*128 + j*16 + r][b = b0 + wb*64 + i*16 + 4g .. +3]
    float *o = p.out + (int64_t)slice * p.csplit;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int a = a0 + wa * 128 + j * 16 + r;
        if (a >= p.Na) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + wb * 64 + i * 16 + g * 4;
            if (b + 3 < p.Nb) *(f32x4 *)(o + (int64_t)a * p.ldo + b) = acc[i][j];
            else
                for (int e = 0; e < 4; ++e)
                    if (b + e < p.Nb) o[(int64_t)a * p.ldo + b + e] = acc[i][j][e];
        }
    }
}

// tile plan of the linear product: 256 x 256 when both extents fill such tiles with <= 7 % more padded area than 128 x 128
int tn_tile(int Na, int Nb) {
    static const int forced = [] { const char *e = getenv("HGR_TN_TILE"); return e ? atoi(e) : 0; }();
    if (forced == 128 || forced == 256) return forced;
    if (Na < 256 || Nb < 256) return 128;
    const int64_t big = (int64_t)((Na + 255) / 256) * ((Nb + 255) / 256) * 65536, small = (int64_t)((Na + 127) / 128) * ((Nb + 127) / 128) * 16384;
    return big * 100 <= small * 107 ? 256 : 128;
}

unsigned magic32(int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

int tn_launch(const void *P, int64_t ldp, const void *Q, int64_t ldq, float *partial, int64_t ldo,
              int M, int Na, int Nb, int kc, int dtype, bool conv, int H, int W, int C, void *stream, const char *name) {
    HGR_REQUIRE(P && Q && partial, "%s: null operand", name);
    HGR_REQUIRE(M >= 1 && Na >= 8 && Nb >= 8 && Na % 8 == 0 && Nb % 8 == 0 && kc >= 64 && kc % 64 == 0, "%s: bad shape M=%d Na=%d Nb=%d kc=%d (Na, Nb multiples of 8; kc of 64)", name, M, Na, Nb, kc);
    HGR_REQUIRE(ldp >= Na && ldp % 8 == 0 && hgr_aligned(P, 16) && hgr_aligned(Q, 16), "%s: operands must be 16-byte aligned with leading dimensions that are multiples of 8", name);
    HGR_REQUIRE(conv || (ldq >= Nb && ldq % 8 == 0), "%s: ldq=%lld must be >= Nb and a multiple of 8", name, (long long)ldq);
    HGR_REQUIRE(ldo >= Nb && ldo % 4 == 0 && hgr_aligned(partial, 16), "%s: partial must be 16-byte aligned, ldo %% 4 == 0, ldo >= Nb", name);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "%s: bad dtype %d", name, dtype);
    const int S = (M + kc - 1) / kc;
    HGR_REQUIRE(S <= 65535, "%s: %d slices exceed the grid limit", name, S);
    TnArgs a;
    a.P = (const char *)P; a.ldp = ldp; a.Q = (const char *)Q; a.ldq = ldq;
    a.out = partial; a.ldo = ldo; a.csplit = (int64_t)Na * ldo;
    a.M = M; a.Na = Na; a.Nb = Nb; a.kc = kc;
    a.tiles_b = (Nb + 127) / 128;
    a.cH = H; a.cW = W; a.cC = C;
    a.mW = conv ? magic32(W) : 0; a.mH = conv ? magic32(H) : 0; a.mC = conv ? magic32(C) : 0;
    hipStream_t s = (hipStream_t)stream;
    if (!conv && tn_tile(Na, Nb) == 256) {
        a.tiles_b = (Nb + 255) / 256;
        dim3 grid256((unsigned)(((Na + 255) / 256) * a.tiles_b * S));
        if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_tn_256<HGR_BF16>), grid256, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((gemm_tn_256<HGR_F16>), grid256, dim3(512), 0, s, a);
        HGR_CHECK_LAUNCH(name);
        return HGR_OK;
    }
    dim3 grid((unsigned)(((Na + 127) / 128) * a.tiles_b), (unsigned)S);
    if (conv) {
        if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_tn_128<HGR_BF16, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_tn_128<HGR_F16, true>), grid, dim3(256), 0, s, a);
    } else {
        if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_tn_128<HGR_BF16, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_tn_128<HGR_F16, false>), grid, dim3(256), 0, s, a);
    }
    HGR_CHECK_LAUNCH(name);
    return HGR_OK;
}

}  // namespace

extern "C" int hgr_gemm_tn_tile(int Na, int Nb) { return tn_tile(Na, Nb); }

extern "C" int hgr_gemm_tn_splitk(const void *P, int64_t ldp, const void *Q, int64_t ldq, float *partial, int64_t ldo,
                                  int M, int Na, int Nb, int kc, int dtype, void *stream) {
    return tn_launch(P, ldp, Q, ldq, partial, ldo, M, Na, Nb, kc, dtype, false, 0, 0, 0, stream, "hgr_gemm_tn_splitk");
}

extern "C" int hgr_conv3x3_wgrad_splitk(const void *dy, int64_t lddy, const void *x, float *partial, int64_t ldo,
                                        int B, int H, int W, int C, int Cout, int kc, int dtype, void *stream) {
    HGR_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 8 && C % 8 == 0 && C <= 16384 && (int64_t)B * H * W < (1ll << 31), "hgr_conv3x3_wgrad_splitk: bad geometry B=%d H=%d W=%d C=%d", B, H, W, C);
    HGR_REQUIRE((int64_t)B * H * W * (W > H ? W : H) < (1ll << 32), "hgr_conv3x3_wgrad_splitk: B*H*W*max(H, W) must stay below 2^32 (pixel index division)");
    return tn_launch(dy, lddy, x, 0, partial, ldo, B * H * W, Cout, 9 * C, kc, dtype, true, H, W, C, stream, "hgr_conv3x3_wgrad_splitk");
}
