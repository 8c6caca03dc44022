// =================================================================================================
// gemm_nt_duo: 256 (M) x 128 (N) x 64 tile, 256 threads = 4 waves as 2 (M) x 2 (N); a wave owns 128 x 64, the register
// tile of gemm_nt_256 (128 accumulators, 4 quadrant phases per K-tile).  TWO workgroups share a CU: 80 KB of LDS and
// <= 256 registers per wave each, one wave of each workgroup per SIMD.
//
// Why: at one workgroup per CU nothing overlaps a tile's prologue (first operand pieces) and epilogue (bias / fp32
// residual loads, stores): ~14 us of a ~35 us tile at K = 768, and a whole round of 256 tiles stores (and, for the
// residual epilogue, re-reads) its 33 - 67 MB in one burst.  A wave's stores and its LDS-DMA loads share one in-order
// vmcnt, so a single persistent workgroup cannot hide them either (profiles/NOTES.md 4.1, finding 3).  Two independent
// workgroups per CU can: while one is in its epilogue or waits for operands, the other one's waves own the matrix
// pipes of the same SIMDs.  No ping-pong groups inside a workgroup, so ONE barrier per phase (4 per K-tile).
//
// LDS: A0 (m-half-0 rows of both wave rows, 16 KB) and A1 (m-half 1) double-buffered, W0 / W1 (n-half 0 / 1 rows of both
// wave columns, 8 KB each) single-buffered: 2 x 32 + 16 = 80 KB.  A piece's slot is refilled in the phase after the
// barrier that follows its last read:
//     ph1 reads W0(t), A0(t)   issues A1(t+1) x4           waits vmcnt(8)  : W1(t) landed
//     ph2 reads W1(t)          issues W0(t+1) x2           waits vmcnt(14) : A1(t) landed
//     ph3 reads A1(t)          issues W1(t+1) x2, A0(t+2) first half x2    (ph4 reads nothing new)
//     ph4                      issues A0(t+2) second half x2   waits vmcnt(6) : W0(t+1), A0(t+1) landed
// (xN = global_load_lds_dwordx4 instructions per thread; the counts are "my N youngest may still be in flight").
// The activations are prefetched 5 - 7 phases ahead, the weights (L2 / MALL resident panels shared by every row panel)
// 3 phases ahead; if a piece is late the partner workgroup's MFMAs fill the gap.
// =================================================================================================
#include "hgr_gemm_common.h"

namespace hgr_gemm {

#ifndef HGR_PAIR_V2
#define HGR_PAIR_V2 1           // 0: the round-5 form of the producer's pair split (A/B builds)
#endif
constexpr int NTD = 256;
constexpr int DUO_A0 = 0, DUO_A1 = 32768, DUO_W0 = 65536, DUO_W1 = 73728, DUO_LDS = 81920;

// 16-byte store that does not keep the line in the XCD's L2 (sc1: write-through, line dropped): a tile's output is never
// re-read by this launch, and 64 tiles in flight per XCD write as many bytes as the L2 holds (experiment: HGR_GEMM_DBG=8)
__device__ __forceinline__ void store16_sc1(void *ptr, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

// One output tile.  MH = 2: the 256 (M) x 128 (N) tile described above.  MH = 1: a HALF tile, 128 x 128 (m-half 0 only: a wave owns
// 64 x 64, two phases per K-tile) for the row panels that do not fill a whole round of the chip's 512 workgroup slots - see
// duo_plan() on the host side and the half-tile schedule below.
template <int DT, int EPI, bool OUT32, int LN, int MH, bool CONV = false>
__device__ __forceinline__ void duo_tile(const GemmArgs &p, char *smem, const int m0, const int n0) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    constexpr int WR = 64 * MH;                    // rows of the tile one wave row owns
    // opaque per call: inside the persistent tile loop every lane-derived offset would otherwise be hoisted out of the loop and kept
    // live across the main loop of every tile (256 registers + up to 94 spilled; with this 220 - 241 as in the one-tile form)
    int tid = threadIdx.x;
    if (LN == 1) asm volatile("" : "+v"(tid));     // (only the producers are compiled with the tile loop)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, g = lane >> 4;

    // per-lane source offsets (bytes from A / W; operands are < 4 GB, checked on the host) of the LDS-DMA instructions of
    // one K-tile: 4 per A piece (32 piece rows each), 2 per W piece.  Piece row pr of A0 = tile row (pr / 64) * 128 + pr % 64,
    // of W0 = tile row (pr / 32) * 64 + pr % 32; A1 / W1 = the same rows + 64 / + 32.  Source chunk ^= row & 7 (rule 21).
    unsigned oA0[4], oA1[4], oW0[2], oW1[2];
    unsigned vA0[CONV ? 4 : 1], vA1[CONV ? 4 : 1];     // CONV: tap-validity masks of the rows behind oA0 / oA1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = (i * 4 + wave) * 64 + lane;
        const int pr = id >> 3, c = (id & 7) ^ (pr & 7);
        const int ra = MH == 2 ? (pr >> 6) * 128 + (pr & 63) : pr;          // half tile: piece row = tile row
        if (CONV) {
            // implicit im2col of a 3 x 3 / pad 1 / stride 1 convolution over NHWC with C % 64 == 0 (a 64-deep K-tile = 64 channels of ONE
            // tap): the lane keeps the (signed) byte offset of tap (0, 0), channel 8 c of its output pixel and a 9-bit mask of the taps
            // inside the image; tap and kernel offset of a K-tile are scalar (issueA)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int gm = min(m0 + ra + half * 64, p.M - 1);
                const int wo = gm % p.cWo, t1 = gm / p.cWo, ho = t1 % p.cHo, b = t1 / p.cHo;
                unsigned vm = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int hi = ho - 1 + t / 3, wi = wo - 1 + t % 3;
                    if (hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW) vm |= 1u << t;
                }
                const int64_t o = ((((int64_t)b * p.cH + ho - 1) * p.cW + wo - 1) * p.cC + c * 8) * 2;
                if (half) { oA1[i] = (unsigned)(int)o; vA1[i] = vm; } else { oA0[i] = (unsigned)(int)o; vA0[i] = vm; }
            }
        } else {
        oA0[i] = (unsigned)(((int64_t)min(m0 + ra, p.M - 1) * p.lda + c * 8) * 2);
        oA1[i] = (unsigned)(((int64_t)min(m0 + ra + 64, p.M - 1) * p.lda + c * 8) * 2);
        }
        if (i < 2) {
            const int rw = (pr >> 5) * 64 + (pr & 31);
            oW0[i] = (unsigned)(((int64_t)min(n0 + rw, p.N - 1) * p.ldw + c * 8) * 2);
            oW1[i] = (unsigned)(((int64_t)min(n0 + rw + 32, p.N - 1) * p.ldw + c * 8) * 2);
        }
    }
    char *const ldsw = smem + wave * 1024;
    const __amdgpu_buffer_rsrc_t rA = dma_rsrc(p.A), rW = dma_rsrc(p.W);
    // halves of an A piece: instructions [2h, 2h + 2)
    // CONV: base of K-tile t = A + kernel offset of its tap + its first channel (scalar); a lane whose tap is outside the image reads zeros
    auto conv_base = [&](int t, int &tap) {
        const int kq0 = t * 64;
        tap = (int)__umulhi((unsigned)kq0, p.cMagic);
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        return p.A + ((int64_t)(ky * p.cW + kx) * p.cC + (kq0 - tap * p.cC)) * 2;
    };
    auto issueA = [&](const unsigned (&off)[4], const unsigned (&vm)[CONV ? 4 : 1], int slot_base, int t, int h) {
        // (HGR_GEMM_DBG, timing experiments only, wrong results: 128 = no A pieces of odd K-tiles - what sharing the A panel between two
        // column tiles would save in LDS-DMA issue; 256 = no A pieces, 512 = no W pieces behind the first two K-tiles)
        if (HGR_LAB_ON(((p.dbg & 128) && (t & 1) && t > 1) || ((p.dbg & 256) && t > 1))) return;
        char *dst = ldsw + slot_base + (t & 1) * 16384;
        if (CONV) {
            int tap;
            const char *base = conv_base(t, tap);
#pragma unroll
            for (int i = 2 * h; i < 2 * h + 2; ++i) {
                const bool ok = (vm[CONV ? i : 0] >> tap) & 1u;
                const char *sp = ok ? base + (int64_t)(int)off[i] : (const char *)hgr_zero_page;
                __builtin_amdgcn_global_load_lds((const AS1 void *)sp, (AS3 void *)(dst + i * 4096), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 2 * h; i < 2 * h + 2; ++i) dma16(rA, p.A, off[i], t * 128, dst + i * 4096);
    };
    auto issueW = [&](const unsigned (&off)[2], int slot_base, int t) {
        if (HGR_LAB_ON((p.dbg & 512) && t > 1)) return;
        char *dst = ldsw + slot_base;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rW, p.W, off[i], t * 128, dst + i * 4096);
    };

    f32x4 acc[MH][2][4][2];     // [m-half][n-half][m tile][n tile]
#pragma unroll
    for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (LN producer, measured and not kept: touching the tile's lines of the residual pair at tile start - one 4-byte LDS-DMA per 64 bytes
    // into an LDS region no operand DMA targets before the prologue's counted wait - so that the epilogue's read-modify-write finds them in
    // L2 / Infinity Cache: out_proj 56.9 -> 66.5 us, c_proj 131 -> 143 us; the prologue waits for them and the epilogue gains nothing.)

    // LN consumer: thread t finalises the statistics of tile row t from the producer's per-slot partial sums.  Called in the
    // epilogue, where its loads travel together with the bias / ln_s / ln_c loads (one exposed round trip per tile, covered
    // by the partner workgroup); at kernel entry it would delay the first LDS-DMA by a memory round trip.
    auto ln_row_stats = [&]() {
        const f32x4 *sp = (const f32x4 *)(p.ln_stats + (int64_t)min(m0 + tid, p.M - 1) * p.ln_slots * 2);
        float s1 = 0.f, s2 = 0.f;
        // the usual row widths get all their loads issued back to back (a load inside a run-time loop is waited for on the spot:
        // six dependent L2 round trips for width 768)
        auto fixed = [&](auto nq_tag) {
            constexpr int NQ = decltype(nq_tag)::value;
            f32x4 t[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) t[i] = sp[i];
#pragma unroll
            for (int i = 0; i < NQ; ++i) { s1 += t[i][0] + t[i][2]; s2 += t[i][1] + t[i][3]; }
        };
        switch (p.ln_slots) {
            case 4: fixed(std::integral_constant<int, 2>()); break;      // width 256
            case 8: fixed(std::integral_constant<int, 4>()); break;      // 512
            case 10: fixed(std::integral_constant<int, 5>()); break;     // 640
            case 12: fixed(std::integral_constant<int, 6>()); break;     // 768
            case 16: fixed(std::integral_constant<int, 8>()); break;     // 1024
            default:
                for (int i = 0; i < p.ln_slots / 2; ++i) { const f32x4 t = sp[i]; s1 += t[0] + t[2]; s2 += t[1] + t[3]; }
        }
        return ln_finalize(s1, s2, 1.0f / (float)p.K, p.ln_eps);
    };
    const int nk = p.K / 64;    // >= 2 (host guarantees)
    const int offA = (wm * 64 + r) * 128;      // + m tile * 2048, within pieces A0 / A1
    const int offW = (wn * 32 + r) * 128;      // + n tile * 2048, within pieces W0 / W1
    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;
    vec8 af[4][2], wf0[2][2], wf1[2][2];
    // Fragment PREFETCH form of the main loop (round 5), compiled into the folded-LayerNorm consumers (LN = 2: c_fc, in_proj, the
    // k / v projection of the last block): measured in the ViT-B/32 step, same box, three interleaved pairs: c_fc 140.4 / 140.1 / 139.3
    // -> 136.9 / 137.0 / 136.5 us, kv 73.5 -> 70.8; the residual producers and the patch GEMM lose 2 - 4 % with it (their epilogue
    // and tail plan want the registers), so they keep the loop below.  -DHGR_DUO_PF=1 compiles it into every instantiation, =0 into none.
#ifndef HGR_DUO_PF
#define HGR_DUO_PF 2
#endif
#ifndef HGR_DUO_PF_DMA_FIRST
#define HGR_DUO_PF_DMA_FIRST 0      // experiment: 1 = a region's LDS-DMAs in front of its MFMAs instead of in their gaps
#endif
    constexpr bool PF_BUILD = HGR_DUO_PF == 1 || (HGR_DUO_PF == 2 && LN == 2);
    const bool pf_tile = MH == 2 && !CONV && PF_BUILD && m0 + 256 <= p.M && n0 + 128 <= p.N && (p.K & 127) == 0 && p.K >= 256;
    if (pf_tile) {
      if constexpr (MH == 2 && !CONV && PF_BUILD) {
        // The main loop of gemm_nt_ws's matrix waves inside the two-workgroup kernel - fragments
        // prefetched ONE PHASE AHEAD into a second register set (Gray-code quadrant walk alternating between even and odd K-tiles: four
        // fragment sets), reads / DMA issue interleaved with the MFMAs, so that no MFMA burst starts with a wait for its own LDS reads.
        //     ph1: Q(X, W0)   prefetch Y(t)                issues W0(t+1) x2, X'(t+2) x4     wait: W1(t) landed      (vmcnt 6)
        //     ph2: Q(Y, W0)   prefetch W1(t)               issues Y'(t+2) x4
        //     ph3: Q(Y, W1)                                issues W1(t+1) x2                 wait: W0(t+1) landed    (vmcnt 10)
        //     ph4: Q(X, W1)   prefetch W0(t+1), X(t+1)
        // (even t: X = A0, Y = A1; odd t: X = A1, Y = A0; X' / Y' = the piece kinds read in ph4(t+1) / ph1(t+2))
        vec8 afB[4][2];
        auto &afA = af;
        auto rdW = [&](vec8 (&wf)[2][2], int base) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf[j][0] = *(const vec8 *)(smem + base + offW + j * 2048 + sw0);
                wf[j][1] = *(const vec8 *)(smem + base + offW + j * 2048 + sw1);
            }
        };
        auto rdA = [&](vec8 (&a)[4][2], int base) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i][0] = *(const vec8 *)(smem + base + offA + i * 2048 + sw0);
                a[i][1] = *(const vec8 *)(smem + base + offA + i * 2048 + sw1);
            }
        };
        auto rdA2 = [&](vec8 (&a)[4][2], int base, int h) {           // m tiles 2 h, 2 h + 1 of a set
#pragma unroll
            for (int i = 2 * h; i < 2 * h + 2; ++i) {
                a[i][0] = *(const vec8 *)(smem + base + offA + i * 2048 + sw0);
                a[i][1] = *(const vec8 *)(smem + base + offA + i * 2048 + sw1);
            }
        };
        // the 8 MFMAs of one k half of a quadrant (independent accumulators; the two halves are separate scheduling regions, so that the
        // scheduler cannot put the two dependent MFMAs of an accumulator back to back)
        auto mmk = [&](f32x4 (&q)[4][2], const vec8 (&wf)[2][2], const vec8 (&a)[4][2], int kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) q[i][j] = T16<DT>::mfma16(wf[j][kk], a[i][kk], q[i][j]);
        };
        // One lane offset per operand: instruction i of a piece reads piece rows 32 i + 8 wave + lane / 8, i.e. tile rows that differ from
        // instruction 0's by a wave-uniform constant (A: 0, 32, 128, 160 rows, + 64 for A1; W: 0, 64 rows, + 32 for W1) - the constant
        // rides in the scalar offset with the K advance (interior tiles: no row clamp).  2 address registers instead of 12.
        const unsigned laneA = oA0[0], laneW = oW0[0];
        const int rowA = (int)p.lda * 2, rowW = (int)p.ldw * 2;
        auto pfA = [&](int half, int slot_base, int tt) {            // half 0 = piece A0 (m-half 0 rows), 1 = A1
            char *dst = ldsw + slot_base + (tt & 1) * 16384;
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(rA, p.A, laneA, tt * 128 + ((i >> 1) * 128 + (i & 1) * 32 + half * 64) * rowA, dst + i * 4096);
        };
        auto pfW = [&](int half, int slot_base, int tt) {
            char *dst = ldsw + slot_base;
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16(rW, p.W, laneW, tt * 128 + (i * 64 + half * 32) * rowW, dst + i * 4096);
        };
        // issue order of the steady state: ... A0(0), A1(0), W0(0) | A1(1), A0(1), W1(0)
        pfA(0, DUO_A0, 0); pfA(1, DUO_A1, 0); pfW(0, DUO_W0, 0);
        pfA(1, DUO_A1, 1); pfA(0, DUO_A0, 1); pfW(1, DUO_W1, 0);
        HGR_RWAIT(10);              // A0(0), A1(0), W0(0) landed
        rdW(wf0, DUO_W0);
        rdA(afA, DUO_A0);
        HGR_RBAR();
        // MODE 0: t + 2 < nk, 1: t = nk - 2, 2: t = nk - 1
        auto kt = [&](int t, auto odd_tag, auto mode_tag) __attribute__((always_inline)) {
            constexpr bool ODD = decltype(odd_tag)::value;
            constexpr int MODE = decltype(mode_tag)::value;
            const int cb = (t & 1) * 16384, nb = ((t + 1) & 1) * 16384;      // this K-tile's / the next one's A buffer
            // every phase = two scheduling regions (k half 0, k half 1): 8 MFMAs each with half of the phase's reads / DMAs in their gaps
            // ---- ph1 ----
            if (MODE <= 1) pfW(0, DUO_W0, t + 1);
            if (ODD) { rdA2(afA, DUO_A0 + cb, 0); mmk(acc[1][0], wf0, afB, 0); } else { rdA2(afB, DUO_A1 + cb, 0); mmk(acc[0][0], wf0, afA, 0); }
            if (MODE <= 1) mfma8_interleave<4, 0, HGR_DUO_PF_DMA_FIRST ? 0 : 2>(); else mfma8_interleave<4, 0, 0>();
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) pfA(ODD ? 1 : 0, ODD ? DUO_A1 : DUO_A0, t + 2);
            if (ODD) { rdA2(afA, DUO_A0 + cb, 1); mmk(acc[1][0], wf0, afB, 1); } else { rdA2(afB, DUO_A1 + cb, 1); mmk(acc[0][0], wf0, afA, 1); }
            if (MODE == 0) mfma8_interleave<4, 0, HGR_DUO_PF_DMA_FIRST ? 0 : 4>(); else mfma8_interleave<4, 0, 0>();
            if (MODE == 0) HGR_RWAIT(6); else if (MODE == 1) HGR_RWAIT(2); else HGR_RWAIT(0);
            // ---- ph2 ----
            rdW(wf1, DUO_W1);
            if (ODD) mmk(acc[0][0], wf0, afA, 0); else mmk(acc[1][0], wf0, afB, 0);
            mfma8_interleave<4, 0, 0>();
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) pfA(ODD ? 0 : 1, ODD ? DUO_A0 : DUO_A1, t + 2);
            if (ODD) mmk(acc[0][0], wf0, afA, 1); else mmk(acc[1][0], wf0, afB, 1);
            if (MODE == 0) mfma8_interleave<0, 0, HGR_DUO_PF_DMA_FIRST ? 0 : 4>();
            HGR_RBAR();
            // ---- ph3 ----
            if (ODD) mmk(acc[0][1], wf1, afA, 0); else mmk(acc[1][1], wf1, afB, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE <= 1) pfW(1, DUO_W1, t + 1);
            if (ODD) mmk(acc[0][1], wf1, afA, 1); else mmk(acc[1][1], wf1, afB, 1);
            if (MODE <= 1) mfma8_interleave<0, 0, HGR_DUO_PF_DMA_FIRST ? 0 : 2>();
            if (MODE == 0) HGR_RWAIT(10); else if (MODE == 1) HGR_RWAIT(2); else HGR_RBAR();
            // ---- ph4 ----
            if (MODE <= 1) { rdW(wf0, DUO_W0); if (ODD) rdA2(afA, DUO_A0 + nb, 0); else rdA2(afB, DUO_A1 + nb, 0); }
            if (ODD) mmk(acc[1][1], wf1, afB, 0); else mmk(acc[0][1], wf1, afA, 0);
            if (MODE <= 1) mfma8_interleave<8, 0, 0>();
            __builtin_amdgcn_sched_barrier(0);
            if (MODE <= 1) { if (ODD) rdA2(afA, DUO_A0 + nb, 1); else rdA2(afB, DUO_A1 + nb, 1); }
            if (ODD) mmk(acc[1][1], wf1, afB, 1); else mmk(acc[0][1], wf1, afA, 1);
            if (MODE <= 1) mfma8_interleave<4, 0, 0>();
            HGR_RBAR();
        };
        // K-tiles in (even, odd) pairs, nk even (checked above): no run-time parity branch inside the loop
        for (int t = 0; t + 2 < nk; t += 2) {
            kt(t, std::false_type(), std::integral_constant<int, 0>());
            kt(t + 1, std::true_type(), std::integral_constant<int, 0>());
        }
        kt(nk - 2, std::false_type(), std::integral_constant<int, 1>());
        kt(nk - 1, std::true_type(), std::integral_constant<int, 2>());
      }
    } else if constexpr (MH == 2) {
        // prologue in steady-state order: A0(0), A1(0), W0(0), W1(0), A0(1)
        issueA(oA0, vA0, DUO_A0, 0, 0); issueA(oA0, vA0, DUO_A0, 0, 1);
        issueA(oA1, vA1, DUO_A1, 0, 0); issueA(oA1, vA1, DUO_A1, 0, 1);
        issueW(oW0, DUO_W0, 0); issueW(oW1, DUO_W1, 0);
        issueA(oA0, vA0, DUO_A0, 1, 0); issueA(oA0, vA0, DUO_A0, 1, 1);
        HGR_RWAIT(6);               // A0(0), W0(0) landed


        // MODE 0: steady state (t + 2 < nk), 1: second-last K-tile, 2: last K-tile
        auto ktile = [&](int t, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            const char *bufA0 = smem + DUO_A0 + (t & 1) * 16384, *bufA1 = smem + DUO_A1 + (t & 1) * 16384;
            // ---- ph1: Q(0,0) ----
    #pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf0[j][0] = *(const vec8 *)(smem + DUO_W0 + offW + j * 2048 + sw0);
                wf0[j][1] = *(const vec8 *)(smem + DUO_W0 + offW + j * 2048 + sw1);
            }
    #pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *(const vec8 *)(bufA0 + offA + i * 2048 + sw0);
                af[i][1] = *(const vec8 *)(bufA0 + offA + i * 2048 + sw1);
            }
            if (MODE <= 1) { issueA(oA1, vA1, DUO_A1, t + 1, 0); issueA(oA1, vA1, DUO_A1, t + 1, 1); }
            if (MODE <= 1) HGR_RWAIT(8); else HGR_RWAIT(0);                 // W1(t) landed
            __builtin_amdgcn_s_setprio(1);
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk)
    #pragma unroll
                for (int i = 0; i < 4; ++i)
    #pragma unroll
                    for (int j = 0; j < 2; ++j) acc[0][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[0][0][i][j]);
            __builtin_amdgcn_s_setprio(0);
            // ---- ph2: Q(0,1) ----
    #pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf1[j][0] = *(const vec8 *)(smem + DUO_W1 + offW + j * 2048 + sw0);
                wf1[j][1] = *(const vec8 *)(smem + DUO_W1 + offW + j * 2048 + sw1);
            }
            if (MODE <= 1) issueW(oW0, DUO_W0, t + 1);
            if (MODE <= 1) HGR_RWAIT(14); else HGR_RBAR();                  // A1(t) landed
            __builtin_amdgcn_s_setprio(1);
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk)
    #pragma unroll
                for (int i = 0; i < 4; ++i)
    #pragma unroll
                    for (int j = 0; j < 2; ++j) acc[0][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[0][1][i][j]);
            __builtin_amdgcn_s_setprio(0);
            // ---- ph3: Q(1,1) ----
    #pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *(const vec8 *)(bufA1 + offA + i * 2048 + sw0);
                af[i][1] = *(const vec8 *)(bufA1 + offA + i * 2048 + sw1);
            }
            if (MODE <= 1) issueW(oW1, DUO_W1, t + 1);
            if (MODE == 0) issueA(oA0, vA0, DUO_A0, t + 2, 0);
            __builtin_amdgcn_sched_barrier(0);                              // no barrier here: nothing is refilled before ph4's barrier that ph3 reads
            __builtin_amdgcn_s_setprio(1);
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk)
    #pragma unroll
                for (int i = 0; i < 4; ++i)
    #pragma unroll
                    for (int j = 0; j < 2; ++j) acc[1][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[1][1][i][j]);
            __builtin_amdgcn_s_setprio(0);
            // ---- ph4: Q(1,0) ----
            if (MODE == 0) issueA(oA0, vA0, DUO_A0, t + 2, 1);
            if (MODE == 0) HGR_RWAIT(6); else if (MODE == 1) HGR_RWAIT(2);  // W0(t+1), A0(t+1) landed
            __builtin_amdgcn_s_setprio(1);
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk)
    #pragma unroll
                for (int i = 0; i < 4; ++i)
    #pragma unroll
                    for (int j = 0; j < 2; ++j) acc[1][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[1][0][i][j]);
            __builtin_amdgcn_s_setprio(0);
        };

        for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>());
        ktile(nk - 2, std::integral_constant<int, 1>());
        ktile(nk - 1, std::integral_constant<int, 2>());
    } else {
        // Half tile (128 x 128): two phases per K-tile, 8 LDS-DMA instructions per thread per K-tile.  LDS: A (16 KB) in THREE
        // buffers at 0 / 16 K / 32 K (the activations come from HBM / Infinity Cache: prefetched two K-tiles ahead), W0 and W1
        // (8 KB each) in TWO buffers each at 48 K.. (L2-resident weight panels, one K-tile ahead): 80 KB like the full tile.
        //     ph1(t) reads W0(t), A(t)   issues W1(t+1) x2, A(t+2) x4   waits vmcnt(12) : W1(t) landed
        //     ph2(t) reads W1(t)         issues W0(t+2) x2              waits vmcnt(8)  : A(t+1), W0(t+1) landed
        // issue order ... W1(t), A(t+1) | W0(t+1) | W1(t+1), A(t+2) | W0(t+2) ...: a count = "my N youngest may still be in flight".
        constexpr int HW0 = 49152, HW1 = 65536;
        auto issueAh = [&](int t, int buf) {
            char *dst = ldsw + buf * 16384;
            if (CONV) {
                int tap;
                const char *base = conv_base(t, tap);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (vA0[CONV ? i : 0] >> tap) & 1u;
                    const char *sp = ok ? base + (int64_t)(int)oA0[i] : (const char *)hgr_zero_page;
                    __builtin_amdgcn_global_load_lds((const AS1 void *)sp, (AS3 void *)(dst + i * 4096), 16, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(rA, p.A, oA0[i], t * 128, dst + i * 4096);
        };
        auto issueWh = [&](const unsigned (&off)[2], int slot_base, int t) {
            char *dst = ldsw + slot_base + (t & 1) * 8192;
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16(rW, p.W, off[i], t * 128, dst + i * 4096);
        };
        // prologue in steady-state order: A(0), W0(0), W1(0), A(1), W0(1)
        issueAh(0, 0); issueWh(oW0, HW0, 0); issueWh(oW1, HW1, 0); issueAh(1, 1); issueWh(oW0, HW0, 1);
        HGR_RWAIT(8);               // A(0), W0(0) landed
        int ab = 0;                 // t % 3
        auto ktile = [&](int t, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;      // 0: steady state (t + 2 < nk), 1: second-last K-tile, 2: last K-tile
            const char *bufA = smem + ab * 16384;
            const char *bufW0 = smem + HW0 + (t & 1) * 8192, *bufW1 = smem + HW1 + (t & 1) * 8192;
            // ---- ph1: Q(0,0) ----
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf0[j][0] = *(const vec8 *)(bufW0 + offW + j * 2048 + sw0);
                wf0[j][1] = *(const vec8 *)(bufW0 + offW + j * 2048 + sw1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *(const vec8 *)(bufA + offA + i * 2048 + sw0);
                af[i][1] = *(const vec8 *)(bufA + offA + i * 2048 + sw1);
            }
            if (MODE <= 1) issueWh(oW1, HW1, t + 1);
            if (MODE == 0) issueAh(t + 2, ab == 0 ? 2 : ab - 1);            // (t + 2) % 3: last read in ph1(t - 1), behind its barrier
            if (MODE == 0) HGR_RWAIT(12); else if (MODE == 1) HGR_RWAIT(8); else HGR_RWAIT(0);      // W1(t) landed
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[0][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[0][0][i][j]);
            __builtin_amdgcn_s_setprio(0);
            // ---- ph2: Q(0,1) ----
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf1[j][0] = *(const vec8 *)(bufW1 + offW + j * 2048 + sw0);
                wf1[j][1] = *(const vec8 *)(bufW1 + offW + j * 2048 + sw1);
            }
            if (MODE == 0) issueWh(oW0, HW0, t + 2);                        // buffer t & 1: W0(t) was read in ph1, behind its barrier
            if (MODE == 0) HGR_RWAIT(8); else if (MODE == 1) HGR_RWAIT(2); else HGR_RBAR();         // A(t+1), W0(t+1) landed
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[0][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[0][1][i][j]);
            __builtin_amdgcn_s_setprio(0);
            ab = ab == 2 ? 0 : ab + 1;
        };
        for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>());
        ktile(nk - 2, std::integral_constant<int, 1>());
        ktile(nk - 1, std::integral_constant<int, 2>());
    }
    HGR_MBAR();                 // every wave's LDS reads are done, no DMA in flight: the staging area is free
    // the epilogue is VALU / LDS work next to the partner workgroup's MFMA clusters (priority 1): run it above them, or its
    // instructions only get the issue slots the matrix stream leaves over (HGR_GEMM_DBG bit 16 = off, for A/B runs)
    if (!HGR_LAB_ON(p.dbg & 16)) __builtin_amdgcn_s_setprio(3);

    // tile (a, b, i, j) of this lane holds C[m][n .. n+3],  m = m0 + wm*128 + a*64 + i*16 + r,  n = n0 + wn*64 + b*32 + j*16 + g*4
    constexpr bool HAS_BIAS = epi_has_bias(EPI);
    constexpr bool HAS_ADD = EPI == HGR_EPI_BIAS_RESIDUAL || EPI == HGR_EPI_ACCUM;
    const bool full = p.vec_ok && m0 + 2 * WR <= p.M && n0 + 128 <= p.N;
    if (full && !OUT32 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (EPI <= HGR_EPI_BIAS_QUICKGELU || EPI == HGR_EPI_BIAS_RELU)) {
        // 16-bit output: the wave's 128 x 64 tile through its private LDS slice (rows of 128 B + 16 B pad), then full
        // 128-byte lines with 16-byte stores (8 lanes per row, 8 rows per instruction)
        constexpr int RS = 144;
        char *my = smem + wave * (128 * RS);
        f32x4 bq[2][2], lsq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                if (LN == 2) { bq[b][j] = *(const f32x4 *)(p.ln_c + n); lsq[b][j] = *(const f32x4 *)(p.ln_s + n); }
                else bq[b][j] = HAS_BIAS ? *(const f32x4 *)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        float2 *lnrow = (float2 *)(smem + 4 * 128 * RS);      // 256 x (mean, rstd) behind the four staging slices
        if (LN == 2) {
            lnrow[tid] = ln_row_stats();
            __syncthreads();
        }
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
        float2 mr = make_float2(0.f, 1.f);
        if (LN == 2) mr = lnrow[wm * WR + a * 64 + i * 16 + r];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (LN == 2) {           // rounding to 16 bits in ONE defined form (ln_out16, hgr_gemm_common.h)
                *(u32x2 *)(my + (a * 64 + i * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 2) = ln_out16<DT, EPI == HGR_EPI_BIAS_QUICKGELU>(mr, acc[a][b][i][j], lsq[b][j], bq[b][j]);
                continue;
            }
            f32x4 v = acc[a][b][i][j] + bq[b][j];
            if (EPI == HGR_EPI_BIAS_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            typename T16<DT>::vec4 o4;
            if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = quick_gelu16<DT>(v[e]);
            } else o4 = cvt4<DT>(v[0], v[1], v[2], v[3]);
            *(typename T16<DT>::vec4 *)(my + (a * 64 + i * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 2) = o4;
        }
        }
        // addresses = wave-uniform 64-bit base + 32-bit per-lane byte offset (one VALU add per store; a 64-bit row * ldc product
        // per access costs ~6 VALU instructions, and this epilogue competes with the partner workgroup for issue slots)
        const int ch = lane & 7, rr = lane >> 3;
        char *cw = (char *)p.C + ((int64_t)(m0 + wm * WR) * p.ldc + n0 + wn * 64) * 2;
        const unsigned ldcB = (unsigned)p.ldc * 2u;
        const unsigned cl = (unsigned)rr * ldcB + ch * 16;
        if (HGR_LAB_ON(p.dbg & 8)) {
#pragma unroll
            for (int q = 0; q < 8 * MH; ++q) store16_sc1(cw + (cl + q * 8 * ldcB), *(const u32x4 *)(my + (q * 8 + rr) * RS + ch * 16));
            return;
        }
        if (LN == 4) {
            // training forward of the MLP: C keeps the pre-activation (backward needs it), ln_xh gets QuickGELU of the ROUNDED
            // pre-activation - the bits hgr_quickgelu16 would produce from C in a second pass, without that pass
            char *gw = (char *)p.ln_xh + ((int64_t)(m0 + wm * WR) * p.ln_ldx + n0 + wn * 64) * 2;
            const unsigned ldgB = (unsigned)p.ln_ldx * 2u;
            const unsigned gl = (unsigned)rr * ldgB + ch * 16;
#pragma unroll
            for (int q = 0; q < 8 * MH; ++q) {
                const u32x4 v = *(const u32x4 *)(my + (q * 8 + rr) * RS + ch * 16);
                *(u32x4 *)(cw + (cl + q * 8 * ldcB)) = v;
                const vec8 h = __builtin_bit_cast(vec8, v);
                vec8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = quick_gelu_train16<DT>((float)h[e]);
                *(u32x4 *)(gw + (gl + q * 8 * ldgB)) = __builtin_bit_cast(u32x4, o);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 8 * MH; ++q)
            *(u32x4 *)(cw + (cl + q * 8 * ldcB)) = *(const u32x4 *)(my + (q * 8 + rr) * RS + ch * 16);
        return;
    }
    if (full && !OUT32 && epi_has_idn16(EPI) && (p.ldc & 7) == 0 && (p.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0) {
        // relu(acc + bias + 16-bit identity) -> 16 bit (bn3(conv3) ; out += identity ; relu, clip/model.py:46-52): the sum is
        // formed in fp32 and rounded once.  4 passes of 32 rows through the wave's LDS slice (fp32 rows of 256 B + 16 B pad);
        // on the way out a lane owns 8 consecutive columns of a row, so the identity is LOADED and the result STORED as
        // 16 bytes per lane over whole 128-byte lines (8 lanes per row, 8 rows per instruction).
        constexpr int RS = 272;
        char *my = smem + wave * (32 * RS);
        f32x4 bq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) bq[b][j] = HAS_BIAS ? *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int r8 = lane >> 3, c8 = lane & 7;
        const int64_t wrow = m0 + wm * WR, wcol = n0 + wn * 64;
        char *cw = (char *)p.C + (wrow * p.ldc + wcol) * 2;
        const char *iw = (const char *)p.res + (wrow * p.ldr + wcol) * 2;
        const unsigned ldcB = (unsigned)p.ldc * 2u, ldiB = (unsigned)p.ldr * 2u;
        const unsigned cl = r8 * ldcB + c8 * 16, il = r8 * ldiB + c8 * 16;
#pragma unroll
        for (int a = 0; a < MH; ++a) {
        float cs[8];                                  // QGELU_GRAD16 + colsum: this lane's 8 columns summed over its rows of the 64-row unit
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int rl = a * 64 + ih * 32;
            u32x4 idn[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) idn[q] = *(const u32x4 *)(iw + (il + (rl + q * 8) * ldiB));
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(f32x4 *)(my + (i2 * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 4) = acc[a][b][ih * 2 + i2][j] + bq[b][j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 lo = *(const f32x4 *)(my + (q * 8 + r8) * RS + c8 * 32);
                const f32x4 hi = *(const f32x4 *)(my + (q * 8 + r8) * RS + c8 * 32 + 16);
                const typename T16<DT>::vec8 iv = __builtin_bit_cast(typename T16<DT>::vec8, idn[q]);
                typename T16<DT>::vec8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (EPI == HGR_EPI_QGELU_GRAD16) {          // dL/dpre = dL/dpost * g'(pre): the hgr_quickgelu16 backward, one rounding
                        o[e] = mul16<DT>(lo[e], quick_gelu_grad((float)iv[e]));
                        o[e + 4] = mul16<DT>(hi[e], quick_gelu_grad((float)iv[e + 4]));
                    } else {
                        o[e] = (E)fmaxf(lo[e] + (float)iv[e], 0.f);
                        o[e + 4] = (E)fmaxf(hi[e] + (float)iv[e + 4], 0.f);
                    }
                }
                *(u32x4 *)(cw + (cl + (rl + q * 8) * ldcB)) = __builtin_bit_cast(u32x4, o);
                if (EPI == HGR_EPI_QGELU_GRAD16 && p.colsum) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[e] += (float)o[e];
                }
            }
        }
        if (EPI == HGR_EPI_QGELU_GRAD16 && p.colsum) {
            // lanes r8 = 0..7 of a column chunk hold disjoint rows: fixed-order butterfly, then one 32-byte row segment per chunk
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cs[e] += __shfl_xor(cs[e], 8); cs[e] += __shfl_xor(cs[e], 16); cs[e] += __shfl_xor(cs[e], 32);
            }
            const int unit = (int)((wrow + a * 64) >> 6);
            if (r8 == 0 && unit < p.colsum_units) {
                float *dst = p.colsum + (int64_t)unit * p.N + wcol + c8 * 8;
                *(f32x4 *)dst = (f32x4){cs[0], cs[1], cs[2], cs[3]};
                *(f32x4 *)(dst + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};
            }
        }
        }
        return;
    }
    if (full && OUT32 && (EPI == HGR_EPI_NONE || EPI == HGR_EPI_BIAS || HAS_ADD) &&
        (LN == 1 || ((p.ldc & 3) == 0 && (!HAS_ADD || EPI == HGR_EPI_ACCUM || (p.ldr & 3) == 0)))) {
        // fp32 output (+ fp32 residual / old C): 4 passes of 32 rows through the wave's private LDS slice (rows of 256 B +
        // 16 B pad); every global access is then 16 bytes per lane over whole 256-byte row segments (2 full lines per row,
        // 4 rows per instruction) instead of 64-byte fragments of 16 rows.  The pass's 8 addend loads are issued before its
        // LDS round trip; residual and C may alias: a pass loads before it stores, and passes touch disjoint rows.
        constexpr int RS = 272;
        char *my = smem + wave * (32 * RS);
        f32x4 bq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bq[b][j] = HAS_BIAS ? *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int rq = lane >> 4, cq = lane & 15;           // row-in-group and 16-byte column chunk of this lane on the way out
        // every global address below = wave-uniform 64-bit base + 32-bit per-lane byte offset (see the 16-bit epilogue)
        const int64_t wrow = m0 + wm * WR, wcol = n0 + wn * 64;
        if (LN == 1) {
            // Producer of a folded LayerNorm.  The residual stream lives in memory as a pair: hi = x rounded to the MFMA type (= the A
            // operand of the next GEMM, no second copy of the stream) and one byte q = the next 8 mantissa bits of x (hgr_common.h:
            // pair_split / pair_dec; 2^-19 |x| with f16 hi, 2^-16 |x| with bf16 hi), far below the 16-bit rounding of every GEMM input.
            // 3 bytes per element each way (rounds 2 - 3: lo = f16(x - hi), 4 bytes).  Plus this wave's 64-column share of the rows' LayerNorm statistics; the
            // 16 reduction chains of a pass (8 row groups x {sum, sum of squares}) advance stage by stage (DPP latencies overlap).
            // 16-byte accesses on the pair (T21 of the CDNA guide: a row-per-lane epilogue is store-ISSUE bound): a lane owns 8
            // consecutive columns of a row (8 lanes per row, 8 rows per instruction), i.e. ONE 16-byte load / store per half of the
            // pair where the first version issued two of 8 bytes: 64 instead of 128 global memory instructions per thread and
            // tile.  The slot sums pair the same values as the 4-column form stage by stage (a lane's 8 columns = the first
            // exchange of two 4-column lanes), so the statistics keep their bits.  Measured against the 8-byte form, same box, interleaved: out 69.1 -> 64.6 us, proj 146 -> 140 us, image tower 6.04 -> 5.89 ms.
            char *hw = (char *)p.ln_xh + (wrow * p.ln_ldx + wcol) * 2;
            char *lw = (char *)p.ln_xl + (wrow * p.ln_ldx + wcol);            // the low byte of the pair: one byte per element
            char *sw = (char *)(p.ln_stats + (wrow * p.ln_slots + (n0 >> 6) + wn) * 2);
            const int r8 = lane >> 3, c8 = lane & 7;
            const unsigned ldxB = (unsigned)p.ln_ldx * 2u, ldsB = (unsigned)p.ln_slots * 8u;
            const unsigned xo = r8 * ldxB + c8 * 16, so = r8 * ldsB;
            const unsigned ldlB = (unsigned)p.ln_ldx, lo8 = r8 * ldlB + c8 * 8;            // byte offsets into the low half (8 columns = 8 bytes per lane)
            // 8 (4 for a half tile) passes of 16 rows; the old pair of pass P + 2 is requested while pass P is worked on (a ring of three
            // 16-register buffers: passes touch disjoint rows, the compiler cannot hoist the loads itself - the pointers alias).  The first
            // version worked in passes of 32 rows with two 32-register buffers: at the 256-register cap of two workgroups per CU hipcc
            // then sent 9 - 16 accumulator quads through scratch in every tile.
            constexpr int NP = MH * 4;
            u32x4 ohb[3][2];
            u32x2 olb[3][2];
            auto pair_load = [&](int buf, int rl) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    ohb[buf][q] = *(const u32x4 *)(hw + (xo + (rl + q * 8) * ldxB));
                    // (HGR_GEMM_DBG bit 64, measurement only: the pair's low half is neither read nor written - the bound of what any
                    // narrower encoding of it can save: 5.12 -> 4.88 ms per step with the 16-bit low half of round 3)
                    if (!HGR_LAB_ON(p.dbg & 64)) olb[buf][q] = *(const u32x2 *)(lw + (lo8 + (rl + q * 8) * ldlB));
                    else olb[buf][q] = (u32x2){0x80808080u, 0x80808080u};
                }
            };
            pair_load(0, 0);
            pair_load(1, 16);
            // the bias of this lane's 8 output columns, added behind the LDS transpose (8 registers; the accumulator layout needs 16)
            const f32x4 b8lo = *(const f32x4 *)(p.bias + wcol + c8 * 8), b8hi = *(const f32x4 *)(p.bias + wcol + c8 * 8 + 4);
            unsigned gbits = 0u;                              // range guard: largest slot sum of squares seen (as bits: inf / NaN rank highest)
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int a = ps >> 2, i = ps & 3;
                const int rl = a * 64 + i * 16;
                const int pb = ps % 3;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *(f32x4 *)(my + r * RS + (b * 32 + j * 16 + g * 4) * 4) = acc[a][b][i][j];
                if (ps + 2 < NP) pair_load((ps + 2) % 3, rl + 32);
                float s1[2], s2[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 lo4 = *(const f32x4 *)(my + (q * 8 + r8) * RS + c8 * 32);
                    const f32x4 hi4 = *(const f32x4 *)(my + (q * 8 + r8) * RS + c8 * 32 + 16);
                    const vec8 oh = __builtin_bit_cast(vec8, ohb[pb][q]);
                    const u32x2 ol = olb[pb][q];
                    float v[8];
#if HGR_PAIR_V2
                    // (acc + bias) + old x (pair_dec4: hi's bits + the low byte's 8 mantissa bits)
                    v[0] = (lo4[0] + b8lo[0]) + pair_dec4<DT, 0>(oh[0], ol[0]); v[1] = (lo4[1] + b8lo[1]) + pair_dec4<DT, 1>(oh[1], ol[0]);
                    v[2] = (lo4[2] + b8lo[2]) + pair_dec4<DT, 2>(oh[2], ol[0]); v[3] = (lo4[3] + b8lo[3]) + pair_dec4<DT, 3>(oh[3], ol[0]);
                    v[4] = (hi4[0] + b8hi[0]) + pair_dec4<DT, 0>(oh[4], ol[1]); v[5] = (hi4[1] + b8hi[1]) + pair_dec4<DT, 1>(oh[5], ol[1]);
                    v[6] = (hi4[2] + b8hi[2]) + pair_dec4<DT, 2>(oh[6], ol[1]); v[7] = (hi4[3] + b8hi[3]) + pair_dec4<DT, 3>(oh[7], ol[1]);
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {              // (acc + bias) + old x (pair_dec: hi's bits + the low byte's 8 mantissa bits)
                        v[e] = (lo4[e] + b8lo[e]) + pair_dec<DT>(oh[e], (ol[0] >> (8 * e)) & 255u);
                        v[e + 4] = (hi4[e] + b8hi[e]) + pair_dec<DT>(oh[e + 4], (ol[1] >> (8 * e)) & 255u);
                    }
#endif
#if HGR_PAIR_V2
                    // pair_split()'s bits through pair_split2 / pair_put_q (hgr_common.h): ~5 instead of ~8 vector instructions per element
                    u32x4 nh;
                    u32x2 nl;
                    {
                        unsigned tq[8];
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) nh[e2] = pair_split2<DT>(v[2 * e2], v[2 * e2 + 1], tq[2 * e2], tq[2 * e2 + 1]);
                        unsigned n0, n1;
                        pair_put_q<DT, 0>(n0, tq[0]); pair_put_q<DT, 1>(n0, tq[1]); pair_put_q<DT, 2>(n0, tq[2]); pair_put_q<DT, 3>(n0, tq[3]);
                        pair_put_q<DT, 0>(n1, tq[4]); pair_put_q<DT, 1>(n1, tq[5]); pair_put_q<DT, 2>(n1, tq[6]); pair_put_q<DT, 3>(n1, tq[7]);
                        nl = (u32x2){n0, n1};
                    }
                    *(u32x4 *)(hw + (xo + (rl + q * 8) * ldxB)) = nh;
#else
                    vec8 nh;
                    u32x2 nl = (u32x2){0u, 0u};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned q0, q1;
                        E h0, h1;
                        pair_split<DT>(v[e], h0, q0);
                        pair_split<DT>(v[e + 4], h1, q1);
                        nh[e] = h0; nh[e + 4] = h1;
                        nl[0] |= q0 << (8 * e);
                        nl[1] |= q1 << (8 * e);
                    }
                    *(u32x4 *)(hw + (xo + (rl + q * 8) * ldxB)) = __builtin_bit_cast(u32x4, nh);
#endif
                    if (!HGR_LAB_ON(p.dbg & 64)) *(u32x2 *)(lw + (lo8 + (rl + q * 8) * ldlB)) = nl;
                    s1[q] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    s2[q] = (__builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3])) +
                            (__builtin_fmaf(v[4], v[4], v[5] * v[5]) + __builtin_fmaf(v[6], v[6], v[7] * v[7]));
                }
                if (HGR_LAB_ON(p.dbg & 32)) continue;
#define HGR_DPP_STAGE8(CTRL) _Pragma("unroll") for (int q = 0; q < 2; ++q) { \
                    s1[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[q]), CTRL, 0xF, 0xF, true)); \
                    s2[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2[q]), CTRL, 0xF, 0xF, true)); }
                HGR_DPP_STAGE8(0xB1) HGR_DPP_STAGE8(0x4E) HGR_DPP_STAGE8(0x141)
#undef HGR_DPP_STAGE8
                if (c8 == 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        *(float2 *)(sw + (so + (rl + q * 8) * ldsB)) = make_float2(s1[q], s2[q]);
                        gbits = max(gbits, __float_as_uint(s2[q]));
                    }
                }
            }
            if (p.ln_flag && gbits > __float_as_uint(p.ln_guard)) atomicMax(p.ln_flag, gbits);
            return;
        }
        const float *addp = EPI == HGR_EPI_ACCUM ? (const float *)p.C : p.res;
        const int64_t ldadd = EPI == HGR_EPI_ACCUM ? p.ldc : p.ldr;
        char *cw = (char *)p.C + (wrow * p.ldc + wcol) * 4;
        const char *aw = (const char *)addp + (wrow * ldadd + wcol) * 4;
        const unsigned ldcB = (unsigned)p.ldc * 4u, ldaB = (unsigned)ldadd * 4u;
        const unsigned cl = rq * ldcB + cq * 16, al = rq * ldaB + cq * 16;
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int rl = a * 64 + ih * 32;                 // first row of the pass inside the wave's 128 rows
            f32x4 ad[8];
            if (HAS_ADD) {
#pragma unroll
                for (int q = 0; q < 8; ++q) ad[q] = *(const f32x4 *)(aw + (al + (rl + q * 4) * ldaB));
            }
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(f32x4 *)(my + (i2 * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 4) = acc[a][b][ih * 2 + i2][j] + bq[b][j];
            f32x4 vq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                vq[q] = *(const f32x4 *)(my + (q * 4 + rq) * RS + cq * 16);
                if (HAS_ADD) vq[q] += ad[q];
            }
            if (HGR_LAB_ON(p.dbg & 8)) {
#pragma unroll
                for (int q = 0; q < 8; ++q) store16_sc1(cw + (cl + (rl + q * 4) * ldcB), __builtin_bit_cast(u32x4, vq[q]));
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) *(f32x4 *)(cw + (cl + (rl + q * 4) * ldcB)) = vq[q];
            }
        }
        return;
    }
    if (LN == 2) {
        // edge tile of an LN consumer (rows beyond M; N is a multiple of 128 by the host's contract): same arithmetic, guarded rows
        float2 *lnrow = (float2 *)smem;
        lnrow[tid] = ln_row_stats();
        __syncthreads();
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * WR + a * 64 + i * 16 + r;
            const float2 mr = lnrow[wm * WR + a * 64 + i * 16 + r];
            if (m >= p.M) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                const f32x4 sq = *(const f32x4 *)(p.ln_s + n), cq4 = *(const f32x4 *)(p.ln_c + n);
                // (N is a multiple of 128 and the rows allow vector access by the host's contract: a whole quad, the same rounding as interior tiles)
                *(u32x2 *)((E *)p.C + (int64_t)m * p.ldc + n) = ln_out16<DT, EPI == HGR_EPI_BIAS_QUICKGELU>(mr, acc[a][b][i][j], sq, cq4);
            }
        }
        return;
    }
    if (LN == 1) {
        // edge tile of an LN producer: the same pass structure with guarded rows (a partial last row panel)
        typedef typename T16<DT>::vec4 hvec4;
        constexpr int RS = 272;
        char *my = smem + wave * (32 * RS);
        const int rq = lane >> 4, cq = lane & 15;
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int row0 = m0 + wm * WR + a * 64 + ih * 32;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(f32x4 *)(my + (i2 * 16 + r) * RS + (b * 32 + j * 16 + g * 4) * 4) =
                    acc[a][b][ih * 2 + i2][j] + *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int64_t row = row0 + q * 4 + rq;
                f32x4 v = *(const f32x4 *)(my + (q * 4 + rq) * RS + cq * 16);
                const bool ok = row < p.M;
                const int64_t at = row * p.ln_ldx + n0 + wn * 64 + cq * 4;
                if (ok) {
                    const hvec4 oh = *(const hvec4 *)((const E *)p.ln_xh + at);
                    const unsigned ol = *(const unsigned *)((const unsigned char *)p.ln_xl + at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += pair_dec<DT>(oh[e], (ol >> (8 * e)) & 255u);
                }
                const float s1 = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
                const float s2 = row16_sum(__builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3]));
                if (ok) {
                    hvec4 nh;
                    unsigned nl = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { unsigned q; E h; pair_split<DT>(v[e], h, q); nh[e] = h; nl |= q << (8 * e); }
                    *(hvec4 *)((E *)p.ln_xh + at) = nh;
                    *(unsigned *)((unsigned char *)p.ln_xl + at) = nl;
                    if (cq == 0) *(float2 *)(p.ln_stats + (row * p.ln_slots + (n0 >> 6) + wn) * 2) = make_float2(s1, s2);
                    if (cq == 0 && p.ln_flag && __float_as_uint(s2) > __float_as_uint(p.ln_guard)) atomicMax(p.ln_flag, __float_as_uint(s2));
                }
            }
        }
        return;
    }
    if (LN == 4) {
        // edge tile of the dual-output forward (rows beyond M; N is a multiple of 128 by the host's contract)
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * WR + a * 64 + i * 16 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                const f32x4 v = acc[a][b][i][j] + *(const f32x4 *)(p.bias + n);
                const typename T16<DT>::vec4 pre = cvt4<DT>(v[0], v[1], v[2], v[3]);
                *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = pre;
                *(typename T16<DT>::vec4 *)((E *)p.ln_xh + (int64_t)m * p.ln_ldx + n) =
                    (typename T16<DT>::vec4){quick_gelu_train16<DT>((float)pre[0]), quick_gelu_train16<DT>((float)pre[1]), quick_gelu_train16<DT>((float)pre[2]), quick_gelu_train16<DT>((float)pre[3])};
            }
        }
        return;
    }
    if (EPI == HGR_EPI_QGELU_GRAD16 && p.colsum) {
        // edge tile (rows beyond M; the host entry guarantees N % 128 == 0 and vector access): the same products and roundings from the
        // accumulator layout, rows >= M contribute nothing; a 16-lane group (r) holds the 16 rows of an m tile -> butterfly over r
#pragma unroll
        for (int a = 0; a < MH; ++a) {
            f32x4 cs[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) cs[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * WR + a * 64 + i * 16 + r;
                const bool ok = m < p.M;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                    if (!ok) continue;
                    const typename T16<DT>::vec4 idn = *(const typename T16<DT>::vec4 *)((const E *)(const void *)p.res + (int64_t)m * p.ldr + n);
                    typename T16<DT>::vec4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = mul16<DT>(acc[a][b][i][j][e], quick_gelu_grad((float)idn[e])); cs[b][j][e] += (float)o[e]; }
                    *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = o;
                }
            }
            const int unit = (m0 + wm * WR + a * 64) >> 6;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = cs[b][j][e];
                    x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
                    cs[b][j][e] = x;
                }
                if (r == 0 && unit < p.colsum_units) *(f32x4 *)(p.colsum + (int64_t)unit * p.N + n0 + wn * 64 + b * 32 + j * 16 + g * 4) = cs[b][j];
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < MH; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * WR + a * 64 + i * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
            if (n < p.N) store_quad<DT, EPI, OUT32>(p, acc[a][b][i][j], m, n);
        }
    }
}

// Block -> tile.  Blocks [0, p.nbig) are full tiles on the first p.big_panels row panels of 256; blocks [p.nbig, gridDim.x) are
// half tiles on the remaining rows (128-row panels): dispatched last, they fill the slots the last, partly empty round of full
// tiles would leave idle with work of half the size.  Each region has its own XCD-aware grouped raster.
template <int DT, int EPI, bool OUT32, int LN = 0, bool CONV = false>
__global__ __launch_bounds__(NTD, 2) void gemm_nt_duo(GemmArgs p) {
    if (p.kc) {                                   // split-K (see gemm_nt_128)
        const int sp = blockIdx.y;
        p.A += (int64_t)sp * p.kc * 2; p.W += (int64_t)sp * p.kc * 2;
        p.C = (char *)p.C + (int64_t)sp * p.csplit * 4;
        p.K = min(p.kc, p.K - sp * p.kc);
    }
    __shared__ __attribute__((aligned(1024))) char smem[DUO_LDS];
    // (measured and not kept: dealing 128 / 256 of the half tiles to the second slot of every CU in the first round, so that a CU's two
    // workgroups are out of phase - producers 62 -> 72-74 us, 131 -> 163 us: half tiles up front only delay the full tiles behind them)
    // Persistent form (p.total > gridDim.x, launch_duo): workgroup b works on the virtual blocks b, b + gridDim.x, ... - the order the
    // dispatcher would have dealt them to its slot.  A workgroup that ends has to see its stores acknowledged before its slot is
    // released and the next one's first operand round trip starts; here the next tile's first LDS-DMAs are in flight beside them.
    // Compiled in for the residual producers (LN == 1) only: every other instantiation keeps its one-tile code, instruction for instruction.
    constexpr bool PERSIST = LN == 1;
    const int total = PERSIST && p.total > 0 ? p.total : (int)gridDim.x;
  auto one_tile = [&](const int vb) {
    const bool half = vb >= p.nbig;
    const int nwg = half ? total - p.nbig : p.nbig;
    const int orig = half ? vb - p.nbig : vb;
    const int tiles_m = half ? p.tiles_m_half : p.big_panels;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int GROUP = p.group;    // default 4: 64 tiles in flight per XCD = 4 row panels of 256 x 16 column panels of 128
    int tm, tn;
    if (p.m_fastest) {
        const int per = GROUP * tiles_m, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, p.tiles_n - first), loc = wg - grp * per;
        tn = first + loc % gs; tm = loc / gs;
    } else {
        const int per = GROUP * p.tiles_n, grp = wg / per, first = grp * GROUP;
        const int gs = min(GROUP, tiles_m - first), loc = wg - grp * per;
        tm = first + loc % gs; tn = loc / gs;
    }
    if (half) duo_tile<DT, EPI, OUT32, LN, 1, CONV>(p, smem, p.big_panels * 256 + tm * 128, tn * 128);
    else duo_tile<DT, EPI, OUT32, LN, 2, CONV>(p, smem, tm * 256, tn * 128);
  };
    if constexpr (PERSIST) {
        for (int vb = (int)blockIdx.x; vb < total; vb += (int)gridDim.x) {
            one_tile(vb);
            // every wave's epilogue has left the LDS staging slices: the next tile's DMAs may land.  A RAW barrier (round 5): __syncthreads()
            // also drains vmcnt here - the workgroup waited for every store of the tile's read-modify-write to be acknowledged before the
            // next tile's first loads were even issued, the opposite of what the persistent form is for
            if (vb + (int)gridDim.x < total) HGR_RBAR();
        }
    } else one_tile((int)blockIdx.x);
}

namespace {
template <int DT>
void launch_duo_dt(const GemmArgs &a, int epi, bool out32, int ln, dim3 grid, hipStream_t s) {
    switch (ln) {
        case 1: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS_RESIDUAL, true, 1>), grid, dim3(NTD), 0, s, a); return;
        case 2:
            if (epi == HGR_EPI_BIAS_QUICKGELU) hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS_QUICKGELU, false, 2>), grid, dim3(NTD), 0, s, a);
            else hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS, false, 2>), grid, dim3(NTD), 0, s, a);
            return;
        case 4: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS, false, 4>), grid, dim3(NTD), 0, s, a); return;
        case 5: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS_RELU, false, 0, true>), grid, dim3(NTD), 0, s, a); return;   // 3 x 3 convolution (implicit im2col loader)
        default: break;
    }
#define HGR_DUO(E) do { if (out32) hipLaunchKernelGGL((gemm_nt_duo<DT, E, true>), grid, dim3(NTD), 0, s, a); \
                        else hipLaunchKernelGGL((gemm_nt_duo<DT, E, false>), grid, dim3(NTD), 0, s, a); } while (0)
    switch (epi) {
        case HGR_EPI_NONE: HGR_DUO(HGR_EPI_NONE); break;
        case HGR_EPI_BIAS: HGR_DUO(HGR_EPI_BIAS); break;
        case HGR_EPI_BIAS_QUICKGELU: HGR_DUO(HGR_EPI_BIAS_QUICKGELU); break;
        case HGR_EPI_BIAS_RELU: HGR_DUO(HGR_EPI_BIAS_RELU); break;
        case HGR_EPI_BIAS_ADD16_RELU: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS_ADD16_RELU, false>), grid, dim3(NTD), 0, s, a); break;
        case HGR_EPI_ACCUM: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_ACCUM, true>), grid, dim3(NTD), 0, s, a); break;
        case HGR_EPI_QGELU_GRAD16: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_QGELU_GRAD16, false>), grid, dim3(NTD), 0, s, a); break;
        default: hipLaunchKernelGGL((gemm_nt_duo<DT, HGR_EPI_BIAS_RESIDUAL, true>), grid, dim3(NTD), 0, s, a); break;
    }
#undef HGR_DUO
}
}  // namespace

// Default: at most two workgroups per CU, each walking its share of the tiles (see gemm_nt_duo); HGR_DUO_PERSIST=0 = one workgroup per
// tile (A/B runs).  Measured on the ViT-B/32 evaluation step, interleaved pairs on one box: every launch persistent 5.064 -> 5.008 ms
// (-1.1 %; by shape proj 128.6 -> 123.5 us, out 50.1 -> 48.5, fc unchanged), the producers only 5.137 -> 5.104 (-0.6 %).  Bit-identical.
static int g_duo_persist = -1;                   // hgr_gemm_set_persist / HGR_DUO_PERSIST
static int duo_persist() {
    if (g_duo_persist < 0) { const char *e = getenv("HGR_DUO_PERSIST"); g_duo_persist = e ? (atoi(e) != 0) : 1; }
    return g_duo_persist;
}
int duo_set_persist(int enabled) { const int prev = duo_persist(); g_duo_persist = enabled; return prev; }
static int duo_slots() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? 2 * pr.multiProcessorCount : 512;
        n &= ~7;                 // whole multiples of the 8 XCDs: virtual block v and physical block v % grid sit on the same XCD
        if (n < 8) n = 8;
    }
    return n;
}

void launch_duo(const GemmArgs &a0, int dtype, int epi, bool out32, int ln, dim3 grid, hipStream_t s) {
    GemmArgs a = a0;
    a.total = 0;
    // ... for the residual producers (ln == 1, the only instantiations compiled with the tile loop) on launches of up to two rounds of
    // the chip's slots - the launches that gain (their epilogue's read-modify-write drains beside the next tile's first operand loads).
    // Launches of many rounds LOSE with the static deal (ViT-L/14 training step 221.2 -> 223.6 ms, RN50 step 9.58 -> 9.66 ms with
    // every launch persistent): the dispatcher's first-free-slot order balances the slots' drifting speeds, the fixed stride does not;
    // the consumer / plain launches of one to two rounds measured neutral to slightly negative (RN50 9.97 -> 10.03 ms).
    const int ds = duo_slots();
    if (duo_persist() && ln == 1 && grid.y == 1 && !a.kc && (int)grid.x > ds && (int)grid.x <= 2 * ds) { a.total = (int)grid.x; grid.x = (unsigned)ds; }
    if (dtype == HGR_BF16) launch_duo_dt<HGR_BF16>(a, epi, out32, ln, grid, s);
    else launch_duo_dt<HGR_F16>(a, epi, out32, ln, grid, s);
}

}  // namespace hgr_gemm
