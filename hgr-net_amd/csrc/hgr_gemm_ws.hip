// =================================================================================================
// gemm_nt_ws: the NT GEMM of the towers with ROLES - 512 threads = 4 matrix waves + 4 helper waves, ONE persistent workgroup
// per CU, 256 (M) x 128 (N) x 64 tiles (the tile, the LDS pieces and the K order per output element of gemm_nt_duo: same bits).
//
// Why (profiles/NOTES.md, round-4 table): in gemm_nt_duo a tile's epilogue (11 us of VALU + 11 us of stores per in_proj launch; the
// residual producers' read-modify-write of the pair: + 17 us on out_proj, + 27 us on c_proj over the bare product) is lost matrix
// time - the partner workgroup alone cannot keep the matrix pipe busy.  A wave's stores and its LDS-DMAs retire through ONE in-order
// vmcnt, so a persistent workgroup cannot overlap them inside one wave either.  Here the work is split by WAVE:
//   * waves 0-3 (one per SIMD, 2 x 2 over the tile, 128 x 64 each = 128 accumulators): ds_read + MFMA, nothing else.  Fragments are
//     prefetched ONE PHASE AHEAD into a second register set, so no MFMA waits for an LDS round trip; the quadrant order alternates
//     between even and odd K-tiles (a Gray-code walk: every phase reuses one operand set of the phase before and finds the other one
//     prefetched) so four fragment sets (96 registers) suffice.  At the end of a tile the accumulators go to LDS in two halves
//     (ds_write_b128, beside the MFMAs of the last phase / ahead of the second phase of the next tile) and the K-tile stream of the
//     next tile simply continues: no prologue, no epilogue, no drain in these waves.
//   * waves 4-7 (the SIMD partners of waves 0-3): issue every LDS-DMA (the counted vmcnt waits are theirs), pick the dumped
//     accumulators up from LDS in row layout (a lane owns 8 consecutive columns of a row: every global access is 16 bytes per lane over
//     whole lines), and run the epilogue - bias / QuickGELU / ReLU, the folded-LayerNorm consumer, the residual producer with its pair
//     read-modify-write and slot statistics - as half passes (8 rows x 4 of a lane's 8 columns) spread over the next tile's phases,
//     UNDER its MFMAs.
// LDS: gemm_nt_duo's 80 KB of operand pieces (A0 / A1 double-buffered, W0 / W1 single) + a 64 KB dump slot + 4 KB of row statistics.
//
// Phase plan (t = K-tile, X = the A piece of ph1 / ph4, Y = the other one; even t: X = A0, odd t: X = A1):
//     ph1: Q(X, W0)   matrix waves prefetch Y(t)                     helpers issue W0(t+1) x2, X'(t+2) x4     wait: W1(t) landed
//     ph2: Q(Y, W0)   prefetch W1(t)                                 issue Y'(t+2) x4
//     ph3: Q(Y, W1)                                                  issue W1(t+1) x2                         wait: W0(t+1) landed
//     ph4: Q(X, W1)   prefetch W0(t+1), X(t+1)
// (X' / Y' = the piece kinds read in ph4(t+1) / ph1(t+2); xN = LDS-DMA instructions per helper thread).  One barrier per phase, all
// eight waves.  A piece is refilled in the phase after the barrier that follows its last read; activations travel 7 phases ahead,
// weights 3.  The helpers' epilogue loads and stores share the vmcnt queue with the DMAs: the counted waits add the number of
// UNCONDITIONAL epilogue instructions issued behind the DMA they wait for (ws_prev_ops / ws_cur_ops: a static table per phase;
// an instruction that is not counted only makes a wait stricter, never unsafe).
//
// What bounds the main loop (ablation builds, HGR_WS_DBG, c_fc shape 25 600 x 3 072 x 768, one MI355X; profiles/NOTES.md round 5):
// everything 123 us; no MFMAs 87 (the LDS-DMA fill alone: 1.47 GB per launch at 17 TB/s, the chip's L2 -> LDS rate); no DMAs 103;
// MFMAs + barriers only 66 (ideal 57); fragment reads + barriers only 51.  A lone matrix wave per SIMD stalls on its own fragment reads
// (~27 cycles each), and the epilogue's vector instructions cost its SIMD's matrix wave the same issue slots whichever wave runs them:
// RESULT - bit-identical to gemm_nt_duo and NOT faster on any tower shape (c_fc 129 -> 136 us, out_proj 42 -> 48.5, c_proj 119 -> 123);
// off by default (HGR_WS / hgr_gemm_set_ws).  Its matrix-wave loop lives on in gemm_nt_duo's LayerNorm consumers (-DHGR_DUO_PF).
// =================================================================================================
#include "hgr_gemm_common.h"

namespace hgr_gemm {

namespace {
constexpr int WS_NT = 512;
constexpr int WS_A0 = 0, WS_A1 = 32768, WS_W0 = 65536, WS_W1 = 73728, WS_DUMP = 81920, WS_LNROW = WS_DUMP + 65536, WS_LDS = WS_LNROW + 4096;
constexpr int WS_TE = 10;     // K-tiles at the start of a tile whose helper code carries epilogue steps (then: the plain loop)

// Unconditional vector-memory instructions of a helper thread in phase g = 4 T + ph (T = K-tile in the tile, ph = 0..3) on top of
// its LDS-DMAs.  prev: epilogue of the PREVIOUS tile (tiles that have one); cur: loads for this tile's own epilogue.
// half passes of phase g = 4 T + ph: ph1 none, ph2 [4 T], ph3 [4 T + 1], ph4 [4 T + 2, 4 T + 3]; half h = pass h / 2, even = first half
constexpr int ws_nhalf(int g) { return g < 0 ? 0 : (g & 3) == 0 ? 0 : (g & 3) == 3 ? 2 : 1; }
constexpr int ws_half0(int g) { return 4 * (g >> 2) + ((g & 3) == 1 ? 0 : (g & 3) == 2 ? 1 : 2); }
constexpr int ws_half_ops(int mode, int h) {
    if (h >= 32) return 0;
    if ((h & 1) == 0) return (mode == WS_LNP && (h >> 1) + 2 <= 15) ? 2 : 0;    // first half: the pair loads of pass P + 2
    return mode == WS_LNP ? 3 : 1;                                               // second half: the stores (pair high / low / slot sums, or C)
}
constexpr int ws_prev_ops(int mode, bool hasb, int g) {
    if (g < 0) return 0;
    if (g == 0) return mode == WS_LNP ? 4 : 0;                                   // pair loads of passes 0, 1
    const int n = ws_nhalf(g), h0 = ws_half0(g);
    return (n >= 1 ? ws_half_ops(mode, h0) : 0) + (n >= 2 ? ws_half_ops(mode, h0 + 1) : 0);
}
// g = 33: the LayerNorm row statistics (LNC); g = 34: the per-column epilogue vectors of THIS tile's columns (bias, or ln_s + ln_c)
constexpr int ws_cur_ops(int mode, bool hasb, int nq, int g) {
    if (g == 33) return mode == WS_LNC ? nq : 0;
    if (g == 34) return mode == WS_LNC ? 4 : (mode == WS_LNP || hasb) ? 2 : 0;
    return 0;
}

template <int N> __device__ __forceinline__ void ws_rwait() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// 16 MFMAs of a phase interleaved with NR fragment reads and NW dump writes (the reads / writes are independent of the MFMAs)
template <int NR, int NW, int M_>
__device__ __forceinline__ void ws_interleave_step() {
    if constexpr (M_ < 16) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int DR = (M_ + 1) * NR / 16 - M_ * NR / 16, DW = (M_ + 1) * NW / 16 - M_ * NW / 16;
        if constexpr (DR > 0) __builtin_amdgcn_sched_group_barrier(0x100, DR, 0);
        if constexpr (DW > 0) __builtin_amdgcn_sched_group_barrier(0x200, DW, 0);
        ws_interleave_step<NR, NW, M_ + 1>();
    }
}
template <int NR, int NW>
__device__ __forceinline__ void ws_interleave() { ws_interleave_step<NR, NW, 0>(); }

// MODE: WS_PLAIN (16-bit C = act(A W^T [+ bias]); ACT 0 none, 1 QuickGELU, 2 ReLU), WS_LNC (folded-LayerNorm consumer, ACT 0 / 1; NQ =
// 16-byte loads per row of slot statistics = K / 128), WS_LNP (residual producer: pair read-modify-write + slot statistics + guard).
// DBG (timing experiments only, wrong results; HGR_WS_DBG): 1 no MFMAs, 2 no LDS-DMAs, 4 no fragment reads, 8 no epilogue stores
template <int DT, int MODE, int ACT, bool HASB, int NQ, int DBG = 0>
__global__ __launch_bounds__(WS_NT) void gemm_nt_ws(GemmArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    __shared__ __attribute__((aligned(1024))) char smem[WS_LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / 64;                      // even, >= WS_TE + 2 (host)
    const int G = (int)gridDim.x, T = p.total;
    const int ntl = (T - (int)blockIdx.x + G - 1) / G;      // tiles of this workgroup: virtual blocks b, b + G, ... (G % 8 == 0: same XCD)

    // virtual block -> tile (gemm_nt_duo's XCD-aware grouped raster)
    auto tile_of = [&](int vb, int &m0, int &n0) {
        const int xcd = vb & 7, q8 = T >> 3, r8 = T & 7;
        const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
        const int GROUP = p.group;
        int tm, tn;
        if (p.m_fastest) {
            const int per = GROUP * p.tiles_m, grp = wg / per, first = grp * GROUP;
            const int gs = min(GROUP, p.tiles_n - first), loc = wg - grp * per;
            tn = first + loc % gs; tm = loc / gs;
        } else {
            const int per = GROUP * p.tiles_n, grp = wg / per, first = grp * GROUP;
            const int gs = min(GROUP, p.tiles_m - first), loc = wg - grp * per;
            tm = first + loc % gs; tn = loc / gs;
        }
        m0 = tm * 256; n0 = tn * 128;
    };

    if (wave < 4) {
        // =========================================== matrix waves ===========================================
        __builtin_amdgcn_s_setprio(2);
        const int wm = wave >> 1, wn = wave & 1;
        const int r = lane & 15, g = lane >> 4;
        const int offA = (wm * 64 + r) * 128, offW = (wn * 32 + r) * 128;
        const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;
        vec8 wf0[2][2] = {}, wf1[2][2] = {}, afA[4][2] = {}, afB[4][2] = {};
        f32x4 acc[2][2][4][2];      // [m-half][n-half][m tile][n tile]
        auto rdW = [&](vec8 (&wf)[2][2], int base) {
            if (DBG & 4) return;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf[j][0] = *(const vec8 *)(smem + base + offW + j * 2048 + sw0);
                wf[j][1] = *(const vec8 *)(smem + base + offW + j * 2048 + sw1);
            }
        };
        auto rdA = [&](vec8 (&af)[4][2], int base) {
            if (DBG & 4) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *(const vec8 *)(smem + base + offA + i * 2048 + sw0);
                af[i][1] = *(const vec8 *)(smem + base + offA + i * 2048 + sw1);
            }
        };
        auto mm = [&](f32x4 (&q)[4][2], const vec8 (&wf)[2][2], const vec8 (&af)[4][2], auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (DBG & 1) { asm volatile("" :: "v"(wf[j][kk]), "v"(af[i][kk])); if (FIRST && kk == 0) q[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                        else q[i][j] = T16<DT>::mfma16(wf[j][kk], af[i][kk], (FIRST && kk == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : q[i][j]);
                    }
        };
        // accumulators of m-half a -> the wave's 16 KB of the dump slot: 64 rows x 256 B (64 fp32 columns), 16-byte chunk ^= row & 15
        auto dump_half = [&](int a) {
            char *dst = smem + WS_DUMP + wave * 16384;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int row = i * 16 + r, chunk = b * 8 + j * 4 + g;
                        *(f32x4 *)(dst + row * 256 + ((chunk ^ (row & 15)) * 16)) = acc[a][b][i][j];
                    }
        };
        auto even_kt = [&](auto first_tag, bool dump1) {
            // ph1: Q(0,0)
            rdA(afB, WS_A1);
            mm(acc[0][0], wf0, afA, first_tag);
            ws_interleave<8, 0>();
            HGR_RBAR();
            // ph2: Q(1,0); the second phase of a tile first hands m-half 1 of the previous tile over
            if (dump1) dump_half(1);
            rdW(wf1, WS_W1);
            mm(acc[1][0], wf0, afB, first_tag);
            HGR_RBAR();
            // ph3: Q(1,1)
            mm(acc[1][1], wf1, afB, first_tag);
            HGR_RBAR();
            // ph4: Q(0,1)
            rdW(wf0, WS_W0);
            rdA(afB, WS_A1 + 16384);
            mm(acc[0][1], wf1, afA, first_tag);
            ws_interleave<12, 0>();
            HGR_RBAR();
        };
        auto odd_kt = [&](auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            // ph1: Q(1,0)
            rdA(afA, WS_A0 + 16384);
            mm(acc[1][0], wf0, afB, std::false_type());
            ws_interleave<8, 0>();
            HGR_RBAR();
            // ph2: Q(0,0)
            rdW(wf1, WS_W1);
            mm(acc[0][0], wf0, afA, std::false_type());
            ws_interleave<4, 0>();
            HGR_RBAR();
            // ph3: Q(0,1)
            mm(acc[0][1], wf1, afA, std::false_type());
            HGR_RBAR();
            // ph4: Q(1,1); the tile's last phase also dumps m-half 0 (final since ph3)
            rdW(wf0, WS_W0);
            rdA(afA, WS_A0);
            if (LAST) dump_half(0);
            mm(acc[1][1], wf1, afB, std::false_type());
            if (LAST) ws_interleave<12, 16>(); else ws_interleave<12, 0>();
            HGR_RBAR();
        };
        HGR_MBAR();                               // P0: A0(0), W0(0) landed
        rdW(wf0, WS_W0);
        rdA(afA, WS_A0);
        HGR_RBAR();                               // P1
        for (int k = 0; k < ntl; ++k) {
            even_kt(std::true_type(), k > 0);
            odd_kt(std::false_type());
            for (int t = 2; t < nk - 2; t += 2) {
                even_kt(std::false_type(), false);
                odd_kt(std::false_type());
            }
            even_kt(std::false_type(), false);
            odd_kt(std::true_type());
        }
        HGR_MBAR();                               // T1: the helpers have picked up m-half 0 of the last tile
        dump_half(1);
        HGR_RBAR();                               // T2
        return;
    }

    // ================================================ helper waves ================================================
    const int lw = wave - 4;
    const int wm = lw >> 1, wn = lw & 1;
    const int ht = tid - 256;                     // 0 .. 255: the tile row whose LayerNorm statistics this thread finalises (WS_LNC)
    const int rr = lane >> 3, ch = lane & 7;      // row-in-group and 8-column chunk of this lane in the row layout
    struct Src { unsigned oA0[4], oA1[4], oW0[2], oW1[2]; int m0, n0; };
    auto set_src = [&](Src &s, int vb) {
        tile_of(vb, s.m0, s.n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = (i * 4 + lw) * 64 + lane;
            const int pr = id >> 3, c = (id & 7) ^ (pr & 7);
            const int ra = (pr >> 6) * 128 + (pr & 63);
            s.oA0[i] = (unsigned)(((int64_t)min(s.m0 + ra, p.M - 1) * p.lda + c * 8) * 2);
            s.oA1[i] = (unsigned)(((int64_t)min(s.m0 + ra + 64, p.M - 1) * p.lda + c * 8) * 2);
            if (i < 2) {
                const int rw = (pr >> 5) * 64 + (pr & 31);
                s.oW0[i] = (unsigned)(((int64_t)min(s.n0 + rw, p.N - 1) * p.ldw + c * 8) * 2);
                s.oW1[i] = (unsigned)(((int64_t)min(s.n0 + rw + 32, p.N - 1) * p.ldw + c * 8) * 2);
            }
        }
    };
    char *const ldsw = smem + lw * 1024;
    const __amdgpu_buffer_rsrc_t rA = dma_rsrc(p.A), rW = dma_rsrc(p.W);
    auto issueA = [&](const unsigned (&off)[4], int dst_base, int kt) {
        if (DBG & 2) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rA, p.A, off[i], kt * 128, ldsw + dst_base + i * 4096);
    };
    auto issueW = [&](const unsigned (&off)[2], int dst_base, int kt) {
        if (DBG & 2) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rW, p.W, off[i], kt * 128, ldsw + dst_base + i * 4096);
    };

    // ---- epilogue state of the tile being handed over (the PREVIOUS tile of the K-tile stream) ----
    f32x4 dlo[8], dhi[8];                         // m-half 0 in row layout: pass q = rows 8 q + rr, this lane's columns 8 ch .. 8 ch + 7
    f32x4 eb[4];                                  // PLAIN / LNP: bias (lo, hi); LNC: ln_s (lo, hi), ln_c (lo, hi) of this lane's 8 columns
    char *cw = nullptr;                           // C (PLAIN / LNC) or xh (LNP) at the wave's first row / column of the tile
    char *lw8 = nullptr, *sw8 = nullptr;          // LNP: the pair's low bytes, the slot statistics
    unsigned ldcB = 0, ldlB = 0, ldsB = 0;
    int ppar = 0;                                 // LNC: which lnrow buffer holds the previous tile's (mean, rstd)
    u32x4 ohb[3];
    u32x2 olb[3];
    unsigned gbits = 0u;
    f32x4 st[NQ > 0 ? NQ : 1];                    // LNC: this thread's row of slot statistics of the CURRENT tile, between its load and its use

    // Every LDS access of the helpers is inline asm with its own lgkmcnt wait: hipcc orders an LDS load / store it can see behind ALL
    // LDS-DMAs still in flight (s_waitcnt vmcnt(0) - it cannot tell the dump slot from the operand pieces), i.e. one full memory
    // round trip per access in a wave that always has DMAs in flight (first version: c_fc 162 us against gemm_nt_duo's 131).
    // Dump slot address of row 8 q + rr: the chunk swizzle depends on q's parity only -> two base pairs + immediate offsets.
    const unsigned dbase = (unsigned)(uintptr_t)(AS3 char *)(smem + WS_DUMP + lw * 16384);
    unsigned dlo0 = dbase + rr * 256 + (((2 * ch) ^ rr) * 16), dhi0 = dbase + rr * 256 + (((2 * ch + 1) ^ rr) * 16);                       // even q
    unsigned dlo1 = dbase + (8 + rr) * 256 + (((2 * ch) ^ (8 + rr)) * 16), dhi1 = dbase + (8 + rr) * 256 + (((2 * ch + 1) ^ (8 + rr)) * 16);   // odd q
    auto dump_read = [&](auto q_tag, f32x4 &lo, f32x4 &hi) {
        constexpr int q = decltype(q_tag)::value;
        const unsigned a0 = (q & 1) ? dlo1 : dlo0, a1 = (q & 1) ? dhi1 : dhi0;
        asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "n"((q >> 1) * 4096) : "memory");
    };
    auto dump_read_all = [&]() {                  // m-half 0 into dlo / dhi: 16 reads, one wait
        asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %17\n\tds_read_b128 %2, %18\n\tds_read_b128 %3, %19\n\t"
                     "ds_read_b128 %4, %16 offset:4096\n\tds_read_b128 %5, %17 offset:4096\n\tds_read_b128 %6, %18 offset:4096\n\tds_read_b128 %7, %19 offset:4096\n\t"
                     "ds_read_b128 %8, %16 offset:8192\n\tds_read_b128 %9, %17 offset:8192\n\tds_read_b128 %10, %18 offset:8192\n\tds_read_b128 %11, %19 offset:8192\n\t"
                     "ds_read_b128 %12, %16 offset:12288\n\tds_read_b128 %13, %17 offset:12288\n\tds_read_b128 %14, %18 offset:12288\n\tds_read_b128 %15, %19 offset:12288\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(dlo[0]), "=&v"(dhi[0]), "=&v"(dlo[1]), "=&v"(dhi[1]), "=&v"(dlo[2]), "=&v"(dhi[2]), "=&v"(dlo[3]), "=&v"(dhi[3]),
                       "=&v"(dlo[4]), "=&v"(dhi[4]), "=&v"(dlo[5]), "=&v"(dhi[5]), "=&v"(dlo[6]), "=&v"(dhi[6]), "=&v"(dlo[7]), "=&v"(dhi[7])
                     : "v"(dlo0), "v"(dhi0), "v"(dlo1), "v"(dhi1) : "memory");
    };
    // LNC: (mean, rstd) of the 16 rows this lane handles (rows wm * 128 + 8 P + rr), from the previous tile's lnrow buffer
    typedef __attribute__((ext_vector_type(2))) float ws_f2;
    ws_f2 mrow[16];
    auto lnrow_read_all = [&]() {
        const unsigned a = (unsigned)(uintptr_t)(AS3 char *)(smem + WS_LNROW) + ppar * 2048 + (wm * 128 + rr) * 8;
        asm volatile("ds_read_b64 %0, %16\n\tds_read_b64 %1, %16 offset:64\n\tds_read_b64 %2, %16 offset:128\n\tds_read_b64 %3, %16 offset:192\n\t"
                     "ds_read_b64 %4, %16 offset:256\n\tds_read_b64 %5, %16 offset:320\n\tds_read_b64 %6, %16 offset:384\n\tds_read_b64 %7, %16 offset:448\n\t"
                     "ds_read_b64 %8, %16 offset:512\n\tds_read_b64 %9, %16 offset:576\n\tds_read_b64 %10, %16 offset:640\n\tds_read_b64 %11, %16 offset:704\n\t"
                     "ds_read_b64 %12, %16 offset:768\n\tds_read_b64 %13, %16 offset:832\n\tds_read_b64 %14, %16 offset:896\n\tds_read_b64 %15, %16 offset:960\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(mrow[0]), "=&v"(mrow[1]), "=&v"(mrow[2]), "=&v"(mrow[3]), "=&v"(mrow[4]), "=&v"(mrow[5]), "=&v"(mrow[6]), "=&v"(mrow[7]),
                       "=&v"(mrow[8]), "=&v"(mrow[9]), "=&v"(mrow[10]), "=&v"(mrow[11]),
                       "=&v"(mrow[12]), "=&v"(mrow[13]), "=&v"(mrow[14]), "=&v"(mrow[15])
                     : "v"(a) : "memory");
    };
    auto pair_load = [&](int P) {                 // LNP: the old pair of pass P (global pass index 0 .. 15: m-half P / 8, rows 8 (P % 8) + rr)
        const unsigned row = (unsigned)(P * 8 + rr);
        ohb[P % 3] = *(const u32x4 *)(cw + (row * ldcB + ch * 16));
        olb[P % 3] = *(const u32x2 *)(lw8 + (row * ldlB + ch * 8));
    };
    // One pass = 8 rows x 64 columns of the wave's tile: row 8 P + rr (P = 0 .. 15 over both m-halves), this lane's 8 columns.  A pass is
    // worked in two HALVES (columns 0-3, then 4-7 + the stores) so that no phase carries more than ~40 vector instructions of it: the
    // helper shares its SIMD's issue slots with a matrix wave, and a whole QuickGELU pass in one phase (first version) stretched that
    // phase by its own length (c_fc: + 18 us per launch over the bias-only epilogue).  State between the halves: kv* below.
    f32x4 khi;                                    // m-half 1: the pass's second dump read, taken with the first
    float ks1 = 0.f, ks2 = 0.f;                   // LNP: the first half's share of the slot sums
    u32x2 knh = {0u, 0u};                         // the first half's four 16-bit results
    unsigned knl = 0u;                            // LNP: ... and low bytes
    auto half_pass = [&](auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
        constexpr int P = h >> 1;
        constexpr bool SECOND = (h & 1) != 0;
        const unsigned row = (unsigned)(P * 8 + rr);
        f32x4 x;
        if constexpr (P < 8) x = SECOND ? dhi[P] : dlo[P];
        else if constexpr (!SECOND) dump_read(std::integral_constant<int, P - 8>(), x, khi);
        else x = khi;
        if constexpr (MODE == WS_LNP) {
            // gemm_nt_duo's LN = 1 epilogue, operation for operation (hgr_gemm_duo.hip): (acc + bias) + old x, split, slot statistics
            const vec8 oh = __builtin_bit_cast(vec8, ohb[P % 3]);
            const unsigned ol = olb[P % 3][SECOND ? 1 : 0];
            float v[4];
            typename T16<DT>::vec4 nh;
            unsigned nl = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (x[e] + eb[SECOND ? 1 : 0][e]) + pair_dec<DT>(oh[e + (SECOND ? 4 : 0)], (ol >> (8 * e)) & 255u);
                unsigned q0;
                E h0;
                pair_split<DT>(v[e], h0, q0);
                nh[e] = h0;
                nl |= q0 << (8 * e);
            }
            const float a1 = (v[0] + v[1]) + (v[2] + v[3]);
            const float a2 = __builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3]);
            if constexpr (!SECOND) { ks1 = a1; ks2 = a2; knh = __builtin_bit_cast(u32x2, nh); knl = nl; }
            else {
                const u32x2 hh = __builtin_bit_cast(u32x2, nh);
                if (DBG & 8) { asm volatile("" :: "v"(hh), "v"(nl)); }
                else {
                    *(u32x4 *)(cw + (row * ldcB + ch * 16)) = (u32x4){knh[0], knh[1], hh[0], hh[1]};
                    *(u32x2 *)(lw8 + (row * ldlB + ch * 8)) = (u32x2){knl, nl};
                }
                float s1 = ks1 + a1, s2 = ks2 + a2;
#define WS_DPP_STAGE(CTRL) \
                s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), CTRL, 0xF, 0xF, true)); \
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), CTRL, 0xF, 0xF, true));
                WS_DPP_STAGE(0xB1) WS_DPP_STAGE(0x4E) WS_DPP_STAGE(0x141)
#undef WS_DPP_STAGE
                if (ch == 0) {
                    *(float2 *)(sw8 + row * ldsB) = make_float2(s1, s2);
                    gbits = max(gbits, __float_as_uint(s2));
                }
            }
        } else {
            u32x2 o2;
            if constexpr (MODE == WS_LNC) o2 = ln_out16<DT, ACT == 1>(make_float2(mrow[P][0], mrow[P][1]), x, eb[SECOND ? 1 : 0], eb[SECOND ? 3 : 2]);   // gemm_nt_duo's rounding
            else {
                typename T16<DT>::vec4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = HASB ? x[e] + eb[SECOND ? 1 : 0][e] : x[e];
                    if (ACT == 2) y = fmaxf(y, 0.f);
                    o[e] = ACT == 1 ? quick_gelu16<DT>(y) : (E)y;
                }
                o2 = __builtin_bit_cast(u32x2, o);
            }
            if constexpr (!SECOND) knh = o2;
            else {
                const u32x2 hh = o2;
                if (DBG & 8) asm volatile("" :: "v"(hh));
                else *(u32x4 *)(cw + (row * ldcB + ch * 16)) = (u32x4){knh[0], knh[1], hh[0], hh[1]};
            }
        }
    };
    // The previous tile's epilogue (origin pm0, pn0) over the phases g = 4 T + ph of this tile: g = 0 set-up + m-half 0 into registers;
    // then per K-tile T = 0 .. 7 the half passes 4 T (ph2), 4 T + 1 (ph3), 4 T + 2 and 4 T + 3 (ph4: the phase without DMAs; ph1, the
    // phase with six, carries none); g = 32 the range guard (LNP).  prev_loads: the global LOADS of a step, issued ahead of the phase's
    // LDS-DMAs (an older load is waited for with a larger count): the pair of pass P + 2 beside the first half of pass P.
    auto prev_loads = [&](auto g_tag, int pm0, int pn0) {
        constexpr int g = decltype(g_tag)::value;
        if constexpr (g == 0) {
            const int64_t wrow = pm0 + wm * 128, wcol = pn0 + wn * 64;
            if constexpr (MODE == WS_LNP) {
                cw = (char *)p.ln_xh + (wrow * p.ln_ldx + wcol) * 2;
                lw8 = (char *)p.ln_xl + (wrow * p.ln_ldx + wcol);
                sw8 = (char *)(p.ln_stats + (wrow * p.ln_slots + (pn0 >> 6) + wn) * 2);
                ldcB = (unsigned)p.ln_ldx * 2u; ldlB = (unsigned)p.ln_ldx; ldsB = (unsigned)p.ln_slots * 8u;
                pair_load(0);
                pair_load(1);
            } else {
                cw = (char *)p.C + (wrow * p.ldc + wcol) * 2;
                ldcB = (unsigned)p.ldc * 2u;
            }
        } else if constexpr (MODE == WS_LNP) {
            constexpr int h0 = ws_half0(g), nh = ws_nhalf(g);
            if constexpr (nh >= 1 && h0 < 32 && (h0 & 1) == 0 && (h0 >> 1) + 2 <= 15) pair_load((h0 >> 1) + 2);
            if constexpr (nh >= 2 && h0 + 1 < 32 && ((h0 + 1) & 1) == 0 && ((h0 + 1) >> 1) + 2 <= 15) pair_load(((h0 + 1) >> 1) + 2);
        }
    };
    auto prev_step = [&](auto g_tag, int pm0, int pn0) {
        constexpr int g = decltype(g_tag)::value;
        if constexpr (g == 0) {
            dump_read_all();
            if constexpr (MODE == WS_LNC) lnrow_read_all();
        } else {
            constexpr int h0 = ws_half0(g), nh = ws_nhalf(g);
            if constexpr (nh >= 1 && h0 < 32) half_pass(std::integral_constant<int, (h0 < 32 ? h0 : 0)>());
            if constexpr (nh >= 2 && h0 + 1 < 32) half_pass(std::integral_constant<int, (h0 + 1 < 32 ? h0 + 1 : 0)>());
            if constexpr (g == 32 && MODE == WS_LNP) {
                // range guard: a conditional atomic - NOT in the counted tables (an extra instruction only makes a wait stricter)
                if (p.ln_flag && gbits > __float_as_uint(p.ln_guard)) atomicMax(p.ln_flag, gbits);
                gbits = 0u;
            }
        }
    };
    // loads for the CURRENT tile's own epilogue: the LayerNorm row statistics (gemm_nt_duo's ln_row_stats) at g = 33, finalised at
    // g = 37; the per-column vectors of this tile's columns at g = 34 (the previous tile's are in use until g = 31)
    auto cur_step = [&](auto g_tag, int m0, int n0, int tpar) {
        constexpr int g = decltype(g_tag)::value;
        if constexpr (g == 34) {
            const int64_t wcol = n0 + wn * 64 + ch * 8;
            if constexpr (MODE == WS_LNC) {
                eb[0] = *(const f32x4 *)(p.ln_s + wcol); eb[1] = *(const f32x4 *)(p.ln_s + wcol + 4);
                eb[2] = *(const f32x4 *)(p.ln_c + wcol); eb[3] = *(const f32x4 *)(p.ln_c + wcol + 4);
            } else if constexpr (MODE == WS_LNP || HASB) {
                eb[0] = *(const f32x4 *)(p.bias + wcol); eb[1] = *(const f32x4 *)(p.bias + wcol + 4);
            }
        }
        if constexpr (MODE == WS_LNC && g == 33) {
            const f32x4 *sp = (const f32x4 *)(p.ln_stats + (int64_t)min(m0 + ht, p.M - 1) * p.ln_slots * 2);
#pragma unroll
            for (int i = 0; i < NQ; ++i) st[i] = sp[i];
        }
        if constexpr (MODE == WS_LNC && g == 37) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NQ; ++i) { s1 += st[i][0] + st[i][2]; s2 += st[i][1] + st[i][3]; }
            const float2 mr = ln_finalize(s1, s2, 1.0f / (float)p.K, p.ln_eps);
            const ws_f2 mv = {mr.x, mr.y};
            asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"((unsigned)(uintptr_t)(AS3 char *)(smem + WS_LNROW) + tpar * 2048 + ht * 8), "v"(mv) : "memory");
        }
    };


    // One K-tile of the helpers.  TT = its index in the tile (>= WS_TE: no epilogue steps), PREV = a previous tile is being handed over.
    // sW / kW: tile and K-tile of the W pieces issued (K-tile t + 1), sA / kA: of the A pieces (t + 2).
    auto kt = [&](auto t_tag, auto prev_tag, const Src &sW, int kW, const Src &sA, int kA, int pm0, int pn0, int m0, int n0, int tpar) {
        constexpr int TT = decltype(t_tag)::value;
        constexpr bool PREV = decltype(prev_tag)::value;
        constexpr bool ODD = (TT & 1) != 0;
        constexpr int N1 = (PREV ? ws_prev_ops(MODE, HASB, 4 * TT - 1) + ws_prev_ops(MODE, HASB, 4 * TT) : 0) + ws_cur_ops(MODE, HASB, NQ, 4 * TT - 1) + ws_cur_ops(MODE, HASB, NQ, 4 * TT) + 6;      // W1(t) was issued first in ph3(t - 1)
        constexpr int N3 = (PREV ? ws_prev_ops(MODE, HASB, 4 * TT + 1) + ws_prev_ops(MODE, HASB, 4 * TT + 2) : 0) + ws_cur_ops(MODE, HASB, NQ, 4 * TT + 1) + ws_cur_ops(MODE, HASB, NQ, 4 * TT + 2) + 10; // W0(t + 1) was issued first in ph1(t)
        auto loads = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            if constexpr (g < 4 * WS_TE && PREV) prev_loads(g_tag, pm0, pn0);
        };
        auto step = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            if constexpr (g < 4 * WS_TE) {
                if constexpr (PREV) prev_step(g_tag, pm0, pn0);
                cur_step(g_tag, m0, n0, tpar);
            }
        };
        // ph1
        loads(std::integral_constant<int, 4 * TT>());
        issueW(sW.oW0, WS_W0, kW);
        issueA(ODD ? sA.oA1 : sA.oA0, ODD ? WS_A1 + 16384 : WS_A0, kA);
        step(std::integral_constant<int, 4 * TT>());
        ws_rwait<N1>();
        // ph2
        loads(std::integral_constant<int, 4 * TT + 1>());
        issueA(ODD ? sA.oA0 : sA.oA1, ODD ? WS_A0 + 16384 : WS_A1, kA);
        step(std::integral_constant<int, 4 * TT + 1>());
        HGR_RBAR();
        // ph3
        loads(std::integral_constant<int, 4 * TT + 2>());
        issueW(sW.oW1, WS_W1, kW);
        step(std::integral_constant<int, 4 * TT + 2>());
        ws_rwait<N3>();
        // ph4
        loads(std::integral_constant<int, 4 * TT + 3>());
        step(std::integral_constant<int, 4 * TT + 3>());
        HGR_RBAR();
    };

    Src cur, nxt;
    set_src(cur, (int)blockIdx.x);
    issueA(cur.oA0, WS_A0, 0); issueA(cur.oA1, WS_A1, 0); issueW(cur.oW0, WS_W0, 0);
    issueA(cur.oA1, WS_A1 + 16384, 1); issueA(cur.oA0, WS_A0 + 16384, 1); issueW(cur.oW1, WS_W1, 0);
    ws_rwait<10>();                               // P0
    HGR_MBAR();                                   // P1
    int pm0 = 0, pn0 = 0;
#define WS_KT(TT, PREV, SW, KW, SA, KA) kt(std::integral_constant<int, TT>(), PREV, SW, KW, SA, KA, pm0, pn0, cur.m0, cur.n0, k & 1)
    for (int k = 0; k < ntl; ++k) {
#define WS_KT10(PREV) WS_KT(0, PREV, cur, 1, cur, 2); WS_KT(1, PREV, cur, 2, cur, 3); WS_KT(2, PREV, cur, 3, cur, 4); WS_KT(3, PREV, cur, 4, cur, 5); \
        WS_KT(4, PREV, cur, 5, cur, 6); WS_KT(5, PREV, cur, 6, cur, 7); WS_KT(6, PREV, cur, 7, cur, 8); WS_KT(7, PREV, cur, 8, cur, 9); \
        WS_KT(8, PREV, cur, 9, cur, 10); WS_KT(9, PREV, cur, 10, cur, 11);
        if (k > 0) { WS_KT10(std::true_type()) } else { WS_KT10(std::false_type()) }
#undef WS_KT10
        for (int t = WS_TE; t < nk - 2; t += 2) {
            WS_KT(WS_TE, std::false_type(), cur, t + 1, cur, t + 2);
            WS_KT(WS_TE + 1, std::false_type(), cur, t + 2, cur, t + 3);
        }
        // the tile's last two K-tiles request the NEXT tile's first pieces (past the last tile: this tile's own, again - nobody reads them)
        if (k + 1 < ntl) set_src(nxt, (int)blockIdx.x + (k + 1) * G); else nxt = cur;
        WS_KT(WS_TE, std::false_type(), cur, nk - 1, nxt, 0);
        WS_KT(WS_TE + 1, std::false_type(), nxt, 0, nxt, 1);
        pm0 = cur.m0; pn0 = cur.n0; ppar = k & 1;
        cur = nxt;
    }
#undef WS_KT
    // the last tile's accumulators: m-half 0 is in the dump slot, m-half 1 follows behind T1
    prev_loads(std::integral_constant<int, 0>(), pm0, pn0);
    prev_step(std::integral_constant<int, 0>(), pm0, pn0);
    HGR_RBAR();                                   // T1 (its lgkmcnt(0): m-half 0 is in registers)
    HGR_MBAR();                                   // T2: m-half 1 is in the dump slot
#define WS_TAIL(G) prev_loads(std::integral_constant<int, G>(), pm0, pn0); prev_step(std::integral_constant<int, G>(), pm0, pn0);
    WS_TAIL(1) WS_TAIL(2) WS_TAIL(3) WS_TAIL(5) WS_TAIL(6) WS_TAIL(7) WS_TAIL(9) WS_TAIL(10) WS_TAIL(11) WS_TAIL(13) WS_TAIL(14) WS_TAIL(15)
    WS_TAIL(17) WS_TAIL(18) WS_TAIL(19) WS_TAIL(21) WS_TAIL(22) WS_TAIL(23) WS_TAIL(25) WS_TAIL(26) WS_TAIL(27) WS_TAIL(29) WS_TAIL(30) WS_TAIL(31) WS_TAIL(32)
#undef WS_TAIL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA of this wave may still be in flight when the workgroup's LDS is released
}
}  // namespace

static int ws_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        n &= ~7;
        if (n < 8) n = 8;
    }
    return n;
}

static int g_ws = -1;                            // hgr_gemm_set_ws / HGR_WS
int ws_enabled() {
    // default OFF: measured equal to or slower than gemm_nt_duo on every tower shape (profiles/NOTES.md, round 5) - the epilogue's
    // vector instructions cost their SIMD's matrix wave the same issue slots whichever wave executes them, and LDS bandwidth bounds both
    if (g_ws < 0) { const char *e = getenv("HGR_WS"); g_ws = e ? (atoi(e) != 0) : 0; }
    return g_ws;
}
int ws_set(int enabled) { const int prev = ws_enabled(); g_ws = enabled; return prev; }

// Whether the role-split kernel covers a launch: whole 256 x 128 tiles, an even number (>= 12) of K-tiles, at least one tile per CU
// (shorter launches keep gemm_nt_duo and its tail plan), 16-bit rows addressable as in gemm_nt_duo
bool ws_covers(int M, int N, int K, int mode) {
    if (M < 256 || M % 256 || N % 128 || K % 128 || K / 64 < WS_TE + 2) return false;
    if ((int64_t)(M / 256) * (N / 128) < ws_cus()) return false;
    if (mode == WS_LNC && !(K == 512 || K == 768 || K == 1024)) return false;
    return true;
}

template <int DT, int MODE, int ACT, bool HASB, int NQ>
static void ws_launch_one(const GemmArgs &a, dim3 g, hipStream_t s) {
    const dim3 b(WS_NT);
#ifdef HGR_LAB                   // ablation builds of the bare product (wrong results): `make lab` only
    static int dbg = -1;
    if (dbg < 0) dbg = hgr_lab_env("HGR_WS_DBG");
    if constexpr (DT == HGR_F16 && MODE == WS_PLAIN && ACT == 0 && !HASB) {
        switch (dbg) {           // ablation builds of the bare product only
            case 1: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 1>), g, b, 0, s, a); return;
            case 2: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 2>), g, b, 0, s, a); return;
            case 3: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 3>), g, b, 0, s, a); return;
            case 4: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 4>), g, b, 0, s, a); return;
            case 6: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 6>), g, b, 0, s, a); return;
            case 7: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 7>), g, b, 0, s, a); return;
            case 8: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 8>), g, b, 0, s, a); return;
            case 15: hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ, 15>), g, b, 0, s, a); return;
            default: break;
        }
    }
#endif
    hipLaunchKernelGGL((gemm_nt_ws<DT, MODE, ACT, HASB, NQ>), g, b, 0, s, a);
}

template <int DT>
static void ws_launch_dt(const GemmArgs &a, int mode, int act, bool hasb, dim3 g, hipStream_t s) {
    if (mode == WS_LNP) { ws_launch_one<DT, WS_LNP, 0, true, 0>(a, g, s); return; }
    if (mode == WS_LNC) {
        const int nq = a.K / 128;
#define WS_LNC_CASE(NQ_) if (nq == NQ_) { if (act) ws_launch_one<DT, WS_LNC, 1, false, NQ_>(a, g, s); else ws_launch_one<DT, WS_LNC, 0, false, NQ_>(a, g, s); return; }
        WS_LNC_CASE(4) WS_LNC_CASE(6) WS_LNC_CASE(8)
#undef WS_LNC_CASE
        return;
    }
    if (!hasb) { ws_launch_one<DT, WS_PLAIN, 0, false, 0>(a, g, s); return; }
    if (act == 2) ws_launch_one<DT, WS_PLAIN, 2, true, 0>(a, g, s);
    else ws_launch_one<DT, WS_PLAIN, 0, true, 0>(a, g, s);
}

// mode WS_PLAIN: act 0 none / 1 QuickGELU / 2 ReLU, hasb = bias; WS_LNC: act 0 / 1; WS_LNP.  The caller checked ws_covers().
void launch_ws(const GemmArgs &a0, int dtype, int mode, int act, bool hasb, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_m = a.M / 256; a.tiles_n = a.N / 128;
    a.total = a.tiles_m * a.tiles_n;
    int grid = ws_cus();
    if (grid > a.total) grid = a.total;
    if (dtype == HGR_BF16) ws_launch_dt<HGR_BF16>(a, mode, act, hasb, dim3((unsigned)grid), s);
    else ws_launch_dt<HGR_F16>(a, mode, act, hasb, dim3((unsigned)grid), s);
}

}  // namespace hgr_gemm
