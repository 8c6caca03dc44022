// =================================================================================================
// gemm_nt_ws: the NT GEMM of the towers with ROLES - 512 threads = 4 matrix waves + 4 helper waves, ONE persistent workgroup
// per CU, 256 (M) x 128 (N) x 64 tiles (the tile, the LDS pieces and the K order per output element of gemm_nt_duo: same bits).
//
// Why (profiles/NOTES.md, round-4 table): in gemm_nt_duo a tile's epilogue (11 us of VALU + 11 us of stores per in_proj launch) is
// lost matrix time - the partner workgroup alone cannot keep the matrix pipe busy, because every wave also issues the LDS-DMAs and
// waits for its own fragment reads.  A wave's stores and its LDS-DMAs retire through ONE in-order vmcnt, so a persistent workgroup
// cannot overlap them either.  Here the work is split by WAVE:
//   * waves 0-3 (one per SIMD, 2 x 2 over the tile, 128 x 64 each = 128 accumulators): ds_read + MFMA, nothing else.  Fragments are
//     prefetched ONE PHASE AHEAD into a second register set, so no MFMA waits for an LDS round trip; the quadrant order alternates
//     between even and odd K-tiles (a Gray-code walk: every phase reuses one operand set of the phase before and finds the other one
//     prefetched) so four fragment sets (96 registers) suffice.  At the end of a tile the accumulators go to LDS in two halves
//     (ds_write_b128, overlapped with the last / the second phase of the neighbouring K-tiles) and the next tile's K-tile stream
//     simply continues.
//   * waves 4-7 (the SIMD partners of waves 0-3): issue every LDS-DMA (the counted vmcnt waits are theirs), pick the dumped
//     accumulators up from LDS in row layout, and run the epilogue + the global stores UNDER the next tile's MFMAs.
// LDS: gemm_nt_duo's 80 KB of operand pieces (A0 / A1 double-buffered, W0 / W1 single) + a 64 KB dump slot = 144 KB.
//
// Phase plan (t = K-tile, X = the A piece of ph1 / ph4, Y = the other one; even t: X = A0, odd t: X = A1):
//     ph1: Q(X, W0)   matrix waves prefetch Y(t)                     helpers issue W0(t+1) x2, X'(t+2) x4     wait: W1(t) landed
//     ph2: Q(Y, W0)   prefetch W1(t)                                 issue Y'(t+2) x4
//     ph3: Q(Y, W1)                                                  issue W1(t+1) x2                         wait: W0(t+1) landed
//     ph4: Q(X, W1)   prefetch W0(t+1), X(t+1)
// (X' / Y' = the piece kinds read in ph4(t+1) / ph1(t+2); xN = LDS-DMA instructions per helper thread).  One barrier per phase, all
// eight waves.  A piece is refilled in the phase after the barrier that follows its last read; activations travel 7 phases ahead,
// weights 3.
// =================================================================================================
#include "hgr_gemm_common.h"

namespace hgr_gemm {

struct WsArgs {
    const char *A; int64_t lda;
    const char *W; int64_t ldw;
    void *C; int64_t ldc;
    const float *bias;
    int M, N, K;
    int tiles_m, tiles_n, m_fastest, group, total;
};

namespace {
constexpr int WS_NT = 512;
constexpr int WS_A0 = 0, WS_A1 = 32768, WS_W0 = 65536, WS_W1 = 73728, WS_DUMP = 81920, WS_LDS = WS_DUMP + 65536;

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

template <int DT, int EPI>
__global__ __launch_bounds__(WS_NT) void gemm_nt_ws(WsArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    __shared__ __attribute__((aligned(1024))) char smem[WS_LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / 64;                      // even, >= 4 (host)
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
        vec8 wf0[2][2], wf1[2][2], afA[4][2], afB[4][2];
        f32x4 acc[2][2][4][2];      // [m-half][n-half][m tile][n tile]
        auto rdW = [&](vec8 (&wf)[2][2], int base) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf[j][0] = *(const vec8 *)(smem + base + offW + j * 2048 + sw0);
                wf[j][1] = *(const vec8 *)(smem + base + offW + j * 2048 + sw1);
            }
        };
        auto rdA = [&](vec8 (&af)[4][2], int base) {
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
                    for (int j = 0; j < 2; ++j)
                        q[i][j] = T16<DT>::mfma16(wf[j][kk], af[i][kk], (FIRST && kk == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : q[i][j]);
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
            // ph2: Q(1,0)
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
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rA, p.A, off[i], kt * 128, ldsw + dst_base + i * 4096);
    };
    auto issueW = [&](const unsigned (&off)[2], int dst_base, int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rW, p.W, off[i], kt * 128, ldsw + dst_base + i * 4096);
    };
    // m-half a of matrix wave lw's accumulators: dump slot -> row layout (a lane owns 8 consecutive columns of a row) -> epilogue -> C
    // as whole 128-byte lines (8 lanes per row, 8 rows per instruction): 16 ds_read_b128 + 8 global stores per lane
    auto drain = [&](int a, int m0, int n0) {
        const int rr = lane >> 3, ch = lane & 7;
        const char *src = smem + WS_DUMP + lw * 16384;
        char *cw = (char *)p.C + ((int64_t)(m0 + wm * 128 + a * 64) * p.ldc + n0 + wn * 64) * 2;
        const unsigned ldcB = (unsigned)p.ldc * 2u;
        f32x4 lo[8], hi[8];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + rr;
            lo[ps] = *(const f32x4 *)(src + row * 256 + (((2 * ch) ^ (row & 15)) * 16));
            hi[ps] = *(const f32x4 *)(src + row * 256 + (((2 * ch + 1) ^ (row & 15)) * 16));
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + rr;
            vec8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (E)lo[ps][e]; o[e + 4] = (E)hi[ps][e]; }
            *(u32x4 *)(cw + ((unsigned)row * ldcB + ch * 16)) = __builtin_bit_cast(u32x4, o);
        }
    };
    // one even + one odd K-tile.  sW / kW: tile and K-tile of the W pieces issued (K-tile t + 1), sA / kA: of the A pieces (t + 2).
    // DUMP: this pair opens a tile whose predecessor's accumulators are picked up in ph1 / ph3 (8 stores each: the counted waits
    // of the following two K-tiles skip them: they are YOUNGER than the DMAs those waits are for)
    auto pair = [&](auto dump_tag, const Src &sW0, int kW0, const Src &sA0, int kA0, const Src &sW1, int kW1, const Src &sA1, int kA1, int pm0, int pn0) {
        constexpr bool DUMP = decltype(dump_tag)::value;
        // ---- even K-tile ----
        issueW(sW0.oW0, WS_W0, kW0); issueA(sA0.oA0, WS_A0, kA0);
        __builtin_amdgcn_sched_barrier(0);        // the DMAs first: the counted waits below rely on the stores being younger
        if (DUMP) drain(0, pm0, pn0);
        if (DUMP) HGR_RWAIT(14); else HGR_RWAIT(6);
        issueA(sA0.oA1, WS_A1, kA0);
        HGR_RBAR();
        issueW(sW0.oW1, WS_W1, kW0);
        __builtin_amdgcn_sched_barrier(0);
        if (DUMP) drain(1, pm0, pn0);
        if (DUMP) HGR_RWAIT(26); else HGR_RWAIT(10);
        HGR_MBAR();
        // ---- odd K-tile ----
        issueW(sW1.oW0, WS_W0, kW1); issueA(sA1.oA1, WS_A1 + 16384, kA1);
        if (DUMP) HGR_RWAIT(14); else HGR_RWAIT(6);
        issueA(sA1.oA0, WS_A0 + 16384, kA1);
        HGR_RBAR();
        issueW(sW1.oW1, WS_W1, kW1);
        HGR_RWAIT(10);
        HGR_MBAR();
    };
    Src cur, nxt;
    set_src(cur, (int)blockIdx.x);
    issueA(cur.oA0, WS_A0, 0); issueA(cur.oA1, WS_A1, 0); issueW(cur.oW0, WS_W0, 0);
    issueA(cur.oA1, WS_A1 + 16384, 1); issueA(cur.oA0, WS_A0 + 16384, 1); issueW(cur.oW1, WS_W1, 0);
    HGR_RWAIT(10);                                // P0
    HGR_MBAR();                                   // P1
    int pm0 = 0, pn0 = 0;
    for (int k = 0; k < ntl; ++k) {
        if (k > 0) pair(std::true_type(), cur, 1, cur, 2, cur, 2, cur, 3, pm0, pn0);
        else pair(std::false_type(), cur, 1, cur, 2, cur, 2, cur, 3, pm0, pn0);
        for (int t = 2; t < nk - 2; t += 2) pair(std::false_type(), cur, t + 1, cur, t + 2, cur, t + 2, cur, t + 3, 0, 0);
        // the tile's last pair requests the NEXT tile's first pieces (past the last tile: this tile's own, again - nobody reads them)
        if (k + 1 < ntl) set_src(nxt, (int)blockIdx.x + (k + 1) * G); else nxt = cur;
        pair(std::false_type(), cur, nk - 1, nxt, 0, nxt, 0, nxt, 1, 0, 0);
        pm0 = cur.m0; pn0 = cur.n0;
        cur = nxt;
    }
    drain(0, pm0, pn0);
    HGR_RBAR();                                   // T1
    HGR_MBAR();                                   // T2
    drain(1, pm0, pn0);
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

void launch_ws(const WsArgs &a0, int dtype, hipStream_t s) {
    WsArgs a = a0;
    a.total = a.tiles_m * a.tiles_n;
    int grid = ws_cus();
    if (grid > a.total) grid = a.total;
    if (dtype == HGR_BF16) hipLaunchKernelGGL((gemm_nt_ws<HGR_BF16, HGR_EPI_NONE>), dim3((unsigned)grid), dim3(WS_NT), 0, s, a);
    else hipLaunchKernelGGL((gemm_nt_ws<HGR_F16, HGR_EPI_NONE>), dim3((unsigned)grid), dim3(WS_NT), 0, s, a);
}

}  // namespace hgr_gemm

using namespace hgr_gemm;

// EXPERIMENT (round 5): C[M, N] (16 bit) = A[M, K] W[N, K]^T with the role-split kernel; M % 256 == 0, N % 128 == 0, K % 128 == 0, K >= 256
extern "C" int hgr_gemm_nt_ws_exp(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                                  int M, int N, int K, int dtype, void *stream) {
    HGR_REQUIRE(A && W && C, "hgr_gemm_nt_ws_exp: null operand");
    HGR_REQUIRE(M >= 256 && M % 256 == 0 && N >= 128 && N % 128 == 0 && K >= 256 && K % 128 == 0, "hgr_gemm_nt_ws_exp: bad shape M=%d N=%d K=%d", M, N, K);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && ldc >= N && ldc % 8 == 0 && ldc < (1 << 23), "hgr_gemm_nt_ws_exp: leading dimensions");
    HGR_REQUIRE(hgr_aligned(A, 16) && hgr_aligned(W, 16) && hgr_aligned(C, 16), "hgr_gemm_nt_ws_exp: operands must be 16-byte aligned");
    HGR_REQUIRE((int64_t)M * lda * 2 < (1ll << 32) && (int64_t)N * ldw * 2 < (1ll << 32), "hgr_gemm_nt_ws_exp: operands beyond 4 GB");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_gemm_nt_ws_exp: bad dtype %d", dtype);
    WsArgs a;
    a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = nullptr;
    a.M = M; a.N = N; a.K = K; a.tiles_m = M / 256; a.tiles_n = N / 128;
    a.m_fastest = ((int64_t)N * K > (int64_t)M * K) ? 1 : 0;
    { static int g = -1; if (g < 0) { const char *e = getenv("HGR_GEMM_GROUP"); g = e ? atoi(e) : 4; if (g < 1) g = 4; } a.group = g; }
    a.total = a.tiles_m * a.tiles_n;
    HGR_REQUIRE(a.total >= 8, "hgr_gemm_nt_ws_exp: fewer than 8 tiles");
    launch_ws(a, dtype, (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_ws_exp");
    return HGR_OK;
}
