// =================================================================================================
// gemm_nt_p8: the LayerNorm-folded consumer GEMM (hgr_gemm_nt_ln: ln_2 -> c_fc -> QuickGELU, clip/model.py:177-187) as ONE persistent
// 512-thread workgroup per CU on 256 x 256 tiles - the main loop of the in_proj + attention kernel (hgr_qkv_attn.hip) with a wider tile
// and a store epilogue in place of the attention phase.
//
// Why a second form of the same product: gemm_nt_duo (256 x 128 tiles, two workgroups per CU) is bound by the L2 -> LDS fill path, not by
// the matrix pipes - the LDS-DMA stream of a CU runs at 66 - 73 GB/s (~33 bytes per clock; MI355X_MICROARCH.md, tools/ldsdma_bw.hip),
// a 256 x 128 tile stages 48 KB per 64-deep K-tile for 64 MFMAs per wave, and with fragment reads, counted waits or barriers compiled
// out the c_fc launch moves by 1 - 5 % (profiles/NOTES.md, round 5).  A 256 x 256 tile stages 64 KB per K-tile for TWICE the MFMAs: 2/3
// of the bytes per flop, fill and matrix time per K-tile about equal (~2 000 clocks each).
//
// 8 waves as 4 (M) x 2 (N); a wave owns 64 x 128 = 4 x 8 MFMA tiles = 128 accumulator registers.  A K-tile is staged as three pieces
// cut by the phase that reads them (PA = the 256 activation rows, 32 KB; PW0 / PW1 = the weight rows of columns 0-63 / 64-127 of every
// wave column, 16 KB each), two stages of 64 KB, every piece refilled for K-tile t + 2 as soon as both ping-pong groups have read it:
//     ph1(t): reads A (8 x ds_read_b128), W columns 0-63 (8 x)    issues PW1(t+1) x2               waits vmcnt(8): PW1(t) landed
//     ph2(t): reads W columns 64-127 (8 x), A fragments stay       issues PA(t+2) x4, PW0(t+2) x2   waits vmcnt(8): PA(t+1), PW0(t+1) landed
// (a phase's LDS-DMAs behind its reads: an LDS-DMA can stall at the CU's one address path; the split by COLUMNS balances the phases'
// LDS time - 16 + 8 reads where a split by rows has 20 + 4 against MFMA bursts of equal length)
// 32 MFMAs per wave and phase; waves 0-3 and 4-7 (one of each per SIMD) run one barrier interval apart.  Same operand roles and K order
// per output element as gemm_nt_duo, hence the same accumulator bits; the epilogue is its consumer epilogue, expression for expression.
//
// Persistent: a workgroup walks its XCD's contiguous range of tiles (column tiles fastest: the tiles in flight on an L2 share row
// panels).  The next tile's first K-tile is requested into stage 0 BEFORE the epilogue, which stages the converted tile through the
// stage-1 region (wave-private 64 x 64 slices, two column halves) so that every global store is 16 bytes per lane over whole lines.
// Every LDS access of the epilogue is inline asm (hipcc orders an LDS access it can see behind ALL LDS-DMAs in flight), and its only
// global loads - row statistics, ln_s / ln_c of the tile's 256 columns, parked in LDS - retire before those DMAs are issued.
// =================================================================================================
#include "hgr_gemm_common.h"

namespace hgr_gemm {
namespace {
constexpr int P8_NT = 512;
constexpr int P8_STAGE = 65536;                              // PA 32 K | PW0 16 K | PW1 16 K
constexpr int P8_PA = 0, P8_PW0 = 32768, P8_PW1 = 49152;
constexpr int P8_RS = 144;                                   // staging row: 64 columns x 2 B + 16 B pad
constexpr int P8_STG = P8_STAGE;                             // 8 waves x 64 rows x 144 B = 73 728 B: stage 1 and the 8 KB behind it
constexpr int P8_LN = P8_STG + 8 * 64 * P8_RS;               // 256 x (mean, rstd)
constexpr int P8_SC = P8_LN + 2048;                          // ln_s[256] | ln_c[256] of the tile's columns
constexpr int P8_LDS = P8_SC + 2048;                         // 143 360 B

typedef __attribute__((ext_vector_type(2))) float p8_f2;

template <int DT, int ACT>
__global__ __launch_bounds__(P8_NT) void gemm_nt_p8(GemmArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    __shared__ __attribute__((aligned(1024))) char smem[P8_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;          // waves w and w + 4 share a SIMD: wn is also the ping-pong group
    const int r = lane & 15, g = lane >> 4;

    // tiles -> workgroups: tile id = row tile * tiles_n + column tile; every XCD (blocks b, b + 8, ... share an L2) owns a contiguous
    // range of ids and its workgroups walk it with a stride of the XCD's workgroup count
    const int ntiles = p.tiles_m * p.tiles_n, orig = blockIdx.x, G = gridDim.x;
    const int xcd = orig & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, xcnt = q8 + (xcd < r8 ? 1 : 0);
    const int xwgs = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    int cur = orig >> 3;
    if (cur >= xcnt) return;

    // per-lane source offsets of one K-tile's LDS-DMA instructions (bytes from A / W; operands < 4 GB, whole tiles: checked on the host)
    unsigned oA[4], oW0[2], oW1[2];
    int m0, n0;
    auto set_tile = [&](int tile) {
        const int tm = tile / p.tiles_n;
        n0 = (tile - tm * p.tiles_n) * 256;
        m0 = tm * 256;
        int ln = lane;
        asm volatile("" : "+v"(ln));                 // opaque: recomputed per tile, not kept live across the main loop
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = (i * 8 + wave) * 64 + ln;
            const int pr = id >> 3, c = (id & 7) ^ (pr & 7);
            oA[i] = (unsigned)(((int64_t)(m0 + pr) * p.lda + c * 8) * 2);          // piece row pr = tile row pr
            if (i < 2) {
                const int tc = (pr >> 6) * 128 + (pr & 63);                        // piece row pr of PW0 = tile column tc; PW1: + 64
                oW0[i] = (unsigned)(((int64_t)(n0 + tc) * p.ldw + c * 8) * 2);
                oW1[i] = (unsigned)(((int64_t)(n0 + tc + 64) * p.ldw + c * 8) * 2);
            }
        }
    };
    set_tile(xbase + cur);
    char *const ldsw = smem + wave * 1024;
    const __amdgpu_buffer_rsrc_t rA = dma_rsrc(p.A), rW = dma_rsrc(p.W);
    auto issueA = [&](int t) {
        char *dst = ldsw + (t & 1) * P8_STAGE + P8_PA;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rA, p.A, oA[i], t * 128, dst + i * 8192);
    };
    auto issueW = [&](const unsigned (&off)[2], int piece, int t) {
        char *dst = ldsw + (t & 1) * P8_STAGE + piece;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rW, p.W, off[i], t * 128, dst + i * 8192);
    };

    const int nk = p.K / 64;    // even, >= 4 (host)
    const int offA = (wm * 64 + r) * 128;          // + m tile (0 .. 3) * 2048 within PA
    const int offW = (wn * 64 + r) * 128;          // + n tile (0 .. 3) * 2048 within PW0 / PW1
    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;
    vec8 wf[4][2], af[4][2];

    // prologue in steady-state order: PA(0), PW0(0) | PW1(0) | PA(1), PW0(1)
    issueA(0); issueW(oW0, P8_PW0, 0);
    issueW(oW1, P8_PW1, 0);
    issueA(1); issueW(oW0, P8_PW0, 1);
  for (;;) {
    f32x4 acc[4][8];            // [m tile][n tile]: C[wm*64 + 16 i + r][wn*128 + 16 j + 4 g .. + 3]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // PA(0), PW0(0) landed: my 8 youngest operations are PW1(0) x2 + PA(1), PW0(1) x6 (first tile), or - across a tile seam, where the
    // previous tile's output stores sit between PW1(0) and PA(1) in issue order - PA(1), PW0(1) x6 and two of those stores
    HGR_RWAIT(8);
    if (wn) HGR_MBAR();         // ping-pong: group 1 runs one barrier interval behind group 0

    // MODE 0: steady state (t + 2 < nk), 1: second-last K-tile, 2: last K-tile
    auto ktile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char *st = smem + (t & 1) * P8_STAGE;
        // ---- ph1: all 64 rows of the wave x its columns 0-63 ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wf[j][0] = *(const vec8 *)(st + P8_PW0 + offW + j * 2048 + sw0);
            wf[j][1] = *(const vec8 *)(st + P8_PW0 + offW + j * 2048 + sw1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const vec8 *)(st + P8_PA + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(st + P8_PA + offA + i * 2048 + sw1);
        }
        if (MODE <= 1) issueW(oW1, P8_PW1, t + 1);              // behind the reads; its slot was last read in ph2(t - 1), two barriers ago
        if (MODE <= 1) HGR_RWAIT(8); else HGR_RWAIT(0);         // PW1(t) landed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T16<DT>::mfma16(wf[j][kk], af[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph2: columns 64-127, the A fragments stay ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wf[j][0] = *(const vec8 *)(st + P8_PW1 + offW + j * 2048 + sw0);
            wf[j][1] = *(const vec8 *)(st + P8_PW1 + offW + j * 2048 + sw1);
        }
        if (MODE == 0) { issueA(t + 2); issueW(oW0, P8_PW0, t + 2); }     // read in ph1(t), two barriers ago
        if (MODE == 0) HGR_RWAIT(8); else if (MODE == 1) HGR_RWAIT(2); else HGR_RBAR();      // PA(t+1), PW0(t+1) landed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][4 + j] = T16<DT>::mfma16(wf[j][kk], af[i][kk], acc[i][4 + j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
    };
    for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>());
    ktile(nk - 2, std::integral_constant<int, 1>());
    ktile(nk - 1, std::integral_constant<int, 2>());
    if (!wn) HGR_MBAR();        // group 0 waits for group 1's last interval: every LDS read is done, no DMA in flight

    const int m0c = m0, n0c = n0;
    const int nxt = cur + xwgs;
    const bool has_next = nxt < xcnt;

    // ---- row statistics of the tile's 256 rows (threads 0-255) and ln_s / ln_c of its 256 columns (threads 256-383) into LDS ----
    const unsigned smem_a = (unsigned)(uintptr_t)(AS3 char *)smem;
    // (opaque copy: every lane-derived term of the epilogue is recomputed per tile instead of being kept live - and spilled - across the main loop)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int le = tid_e & 63, re = le & 15, ge = le >> 4;
    if (tid_e < 256) {
        const f32x4 *sp = (const f32x4 *)(p.ln_stats + (int64_t)(m0c + tid_e) * p.ln_slots * 2);
        float s1 = 0.f, s2 = 0.f;
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
        const float2 mr = ln_finalize(s1, s2, 1.0f / (float)p.K, p.ln_eps);
        const p8_f2 mv = {mr.x, mr.y};
        asm volatile("ds_write_b64 %0, %1" ::"v"(smem_a + P8_LN + tid_e * 8), "v"(mv) : "memory");
    } else if (tid_e < 384) {
        const int q = tid_e - 256;                   // 0-63: ln_s, 64-127: ln_c; four columns each
        const f32x4 v = *(const f32x4 *)((q < 64 ? p.ln_s : p.ln_c) + n0c + (q & 63) * 4);
        asm volatile("ds_write_b128 %0, %1" ::"v"(smem_a + P8_SC + q * 16), "v"(v) : "memory");
    }
    HGR_RBAR();
    if (has_next) {
        // (persistent) the next tile's first K-tile into stage 0, which the epilogue does not touch - behind the epilogue's last global load
        __builtin_amdgcn_sched_barrier(0);
        set_tile(xbase + nxt);
        issueA(0); issueW(oW0, P8_PW0, 0);
        issueW(oW1, P8_PW1, 0);
    }

    // ---- y = act(rstd (acc - mean s_n) + c_n), rounded to the MFMA type, through the wave's staging slice, two column halves ----
    p8_f2 mrv[4];
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:128\n\tds_read_b64 %2, %4 offset:256\n\tds_read_b64 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(mrv[0]), "=&v"(mrv[1]), "=&v"(mrv[2]), "=&v"(mrv[3]) : "v"(smem_a + P8_LN + (wm * 64 + re) * 8) : "memory");
    const unsigned stg = smem_a + P8_STG + wave * (64 * P8_RS);
    const int ch = le & 7, rr = le >> 3;
    const unsigned ldcB = (unsigned)p.ldc * 2u;
    const unsigned cl = (unsigned)rr * ldcB + ch * 16;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = half * 4 + jj;
            f32x4 sv, cv;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(sv), "=&v"(cv) : "v"(smem_a + P8_SC + (wn * 128 + j * 16 + ge * 4) * 4) : "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 mri = make_float2(mrv[i][0], mrv[i][1]);
                const u32x2 pk = ln_out16<DT, ACT != 0>(mri, acc[i][j], sv, cv);
                asm volatile("ds_write_b64 %0, %1" ::"v"(stg + (i * 16 + re) * P8_RS + (jj * 16 + ge * 4) * 2), "v"(pk) : "memory");
            }
        }
        // the wave's own writes, read back as whole 128-byte row halves (8 lanes per row, 8 rows per instruction): DS operations of one wave
        // execute in order
        u32x4 o[8];
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1152\n\tds_read_b128 %2, %8 offset:2304\n\tds_read_b128 %3, %8 offset:3456\n\t"
                     "ds_read_b128 %4, %8 offset:4608\n\tds_read_b128 %5, %8 offset:5760\n\tds_read_b128 %6, %8 offset:6912\n\tds_read_b128 %7, %8 offset:8064\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                     : "v"(stg + rr * P8_RS + ch * 16) : "memory");
        char *cw = (char *)p.C + ((int64_t)(m0c + wm * 64) * p.ldc + n0c + wn * 128 + half * 64) * 2;
#pragma unroll
        for (int q = 0; q < 8; ++q) *(u32x4 *)(cw + (cl + q * 8 * ldcB)) = o[q];
    }
    HGR_RBAR();                 // every wave has left the staging slices: stage 1 may be refilled
    if (!has_next) break;
    issueA(1); issueW(oW0, P8_PW0, 1);
    cur = nxt;
  }
}

int g_p8 = -1;                  // hgr_gemm_set_p8 / HGR_P8: 0 never, 1 wherever p8_covers, 2 by shape (p8_wanted; the default)
int p8_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return n;
}
}  // namespace

int p8_mode() {
    if (g_p8 < 0) { const char *e = getenv("HGR_P8"); const int v = e ? atoi(e) : 2; g_p8 = v >= 0 && v <= 2 ? v : 2; }
    return g_p8;
}
int p8_set(int mode) { const int prev = p8_mode(); g_p8 = mode; return prev; }

// By shape: measured against gemm_nt_duo (tools/p8_bench.py, back to back, one MI355X, bit-identical): 25 600 x 3 072 x 3 072
// 436 -> 394 us (1 108 -> 1 228 TF/s), ViT-L/14's c_fc 65 536 x 4 096 x 1 024 543 -> 519, ViT-B/32's c_fc (K = 768, 4.7 tiles per CU)
// 128.0 -> 126.2 back to back and 132.5 -> 131.8 in the step (the step itself unchanged), its k / v projection (2.3 tiles per CU)
// 70.9 -> 74.3: a K-tile of this form takes ~3 400 clocks where fill, matrix and LDS time are ~2 000 each (ablation builds at
// K = 3 072: 390 us; no fragment reads 378; no LDS-DMA behind the prologue 318; neither 274 - the fill that does not overlap is
// added), and a tile's prologue and epilogue run beside nothing - so it pays with long K and many tiles per CU only.
bool p8_wanted(int M, int N, int K) {
    const int m = p8_mode();
    if (!m || !p8_covers(M, N, K)) return false;
    return m == 1 || (K >= 1024 && (int64_t)(M / 256) * (N / 256) >= 4 * p8_cus());
}

// whole 256 x 256 tiles, an even number of K-tiles, at least one tile per CU (fewer: gemm_nt_duo's 512 slots fill the chip better)
bool p8_covers(int M, int N, int K) {
    return M % 256 == 0 && N % 256 == 0 && K % 128 == 0 && K >= 256 && (int64_t)(M / 256) * (N / 256) >= p8_cus();
}

void launch_p8(const GemmArgs &a0, int dtype, int act, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_m = a.M / 256; a.tiles_n = a.N / 256;
    const int tiles = a.tiles_m * a.tiles_n;
    const dim3 grid((unsigned)(tiles < p8_cus() ? tiles : p8_cus())), block(P8_NT);
    if (dtype == HGR_BF16) {
        if (act) hipLaunchKernelGGL((gemm_nt_p8<HGR_BF16, 1>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_p8<HGR_BF16, 0>), grid, block, 0, s, a);
    } else {
        if (act) hipLaunchKernelGGL((gemm_nt_p8<HGR_F16, 1>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_p8<HGR_F16, 0>), grid, block, 0, s, a);
    }
}

}  // namespace hgr_gemm
