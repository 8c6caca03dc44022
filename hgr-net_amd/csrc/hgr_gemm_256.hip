// =================================================================================================
// 256 x 256 x 64 tile, 512 threads = 8 waves as 2 (M) x 4 (N); each wave owns 128 (M) x 64 (N) =
// 8 x 4 MFMA tiles = 128 accumulator registers.  128 flop per staged byte (2x the 128^2 kernel), which
// is what the short-K tower GEMMs (K = 768: only 12 K-tiles) need: they are bound by bytes in flight.
//
// A K-tile is computed in 4 phases of 16 MFMAs per wave (one quadrant of the wave's output each):
//     ph1: Q(m-half 0, n-half 0)   reads W n-half 0 (4 x ds_read_b128) and A m-half 0 (8 x)
//     ph2: Q(0, 1)                 reads W n-half 1 (4 x)
//     ph3: Q(1, 1)                 reads A m-half 1 (8 x, same registers as m-half 0)
//     ph4: Q(1, 0)                 reads nothing (W n-half 0 is still in registers)
// The K-tile is staged as 4 LDS-DMA "pieces" of 16 KB cut BY PHASE, not by row range:
//     A0 = the m-half-0 rows of both wave rows (A rows 0-63,128-191)   first read in ph1
//     W0 = the n-half-0 rows of all four wave columns                  first read in ph1
//     W1 = the n-half-1 rows                                           first read in ph2
//     A1 = the m-half-1 rows (A rows 64-127,192-255)                   first read in ph3
// so a piece's LDS slot is free one phase after its first read and can be refilled for K-tile t+2
// immediately: every phase issues exactly one piece (2 x global_load_lds_dwordx4 per thread) about
// 6 phases before it is read, and 5 pieces (80 KB per CU) stay in flight across every wait:
//     (t,1) issues A1(t+1)   (t,2) A0(t+2)   (t,3) W0(t+2)   (t,4) W1(t+2)
// Waits are counted: vmcnt(10) = "all but my 10 youngest DMA instructions landed" = the piece the
// NEXT phase reads has landed; then one raw s_barrier (every wave's share landed / every wave's reads
// of the slot about to be refilled are done).  The last two K-tiles use the exact smaller counts.
// Two LDS buffers x 4 pieces = 128 KB, one workgroup per CU.  (cdna_hip_programming.md section 5:
// "Pipelining across barriers", counted vmcnt + raw s_barrier, all LDS in one array.)
// =================================================================================================
#include "hgr_gemm_common.h"

namespace hgr_gemm {

constexpr int NT256 = 512;
constexpr int PIECE = 16384;

template <int DT, int EPI, bool OUT32, bool CONV = false>
__global__ __launch_bounds__(NT256) void gemm_nt_256(GemmArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::elem E;
    if (p.kc) {                                   // split-K (see gemm_nt_128)
        const int sp = blockIdx.y;
        p.A += (int64_t)sp * p.kc * 2; p.W += (int64_t)sp * p.kc * 2;
        p.C = (char *)p.C + (int64_t)sp * p.csplit * 4;
        p.K = min(p.kc, p.K - sp * p.kc);
    }
    __shared__ __attribute__((aligned(1024))) char smem[2 * 4 * PIECE + 8 * 2048];   // + epilogue row padding

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;      // waves w and w+4 share a SIMD: wm is also the ping-pong group
    const int r = lane & 15, g = lane >> 4;

    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    constexpr int GROUP = 4;      // 32 tiles in flight per XCD = 4 x 8 panels of 256 x K
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
    const int m0 = tm * 256, n0 = tn * 256;

    // piece kinds in issue order: 0 = A0, 1 = W0, 2 = W1, 3 = A1.  Each thread moves 2 x 16 B per piece.
    const char *src[4][2];
    int cchunk[2];              // CONV: this lane's logical 16-B chunk (8 input channels) inside a K-tile
    unsigned vmask[2][2];       // CONV: [A0 / A1][j]: bit t set = tap t of that output pixel is inside the image
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int id = (j * 8 + wave) * 64 + lane;       // 16-B chunk id inside the 128-row piece
        const int pr = id >> 3, c = (id & 7) ^ (pr & 7); // piece row, swizzled source chunk
        const int ra0 = pr + (pr >= 64 ? 64 : 0);        // A row of piece row (m-half 0 of wave row pr/64)
        const int rw0 = (pr >> 5) * 64 + (pr & 31);      // W row of piece row (n-half 0 of wave col pr/32)
        src[1][j] = p.W + ((int64_t)min(n0 + rw0, p.N - 1) * p.ldw + c * 8) * 2;
        src[2][j] = p.W + ((int64_t)min(n0 + rw0 + 32, p.N - 1) * p.ldw + c * 8) * 2;
        if (CONV) {
            cchunk[j] = c;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int gm = min(m0 + ra0 + half * 64, p.M - 1);
                const int wo = gm % p.cWo, t1 = gm / p.cWo, ho = t1 % p.cHo, b = t1 / p.cHo;
                const int hi0 = ho * p.cStride - 1, wi0 = wo * p.cStride - 1;
                unsigned vm = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int hi = hi0 + t / 3, wi = wi0 + t % 3;
                    if (hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW) vm |= 1u << t;
                }
                vmask[half][j] = vm;
                src[half ? 3 : 0][j] = p.A + (((int64_t)b * p.cH + hi0) * p.cW + wi0) * p.cC * 2;   // tap (0,0), channel 0
                if (p.cUni) src[half ? 3 : 0][j] += c * 16;                                         // ... + this lane's chunk (uniform-tap path)
            }
        } else {
            src[0][j] = p.A + ((int64_t)min(m0 + ra0, p.M - 1) * p.lda + c * 8) * 2;
            src[3][j] = p.A + ((int64_t)min(m0 + ra0 + 64, p.M - 1) * p.lda + c * 8) * 2;
        }
    }
    const bool do_mma = !HGR_LAB_ON(p.dbg == 1), do_ld = !HGR_LAB_ON(p.dbg == 2);
    auto issue = [&](int kind, int t) {
        if (!do_ld) return;
        char *dst = smem + (t & 1) * (4 * PIECE) + kind * PIECE + wave * 1024;
        const int64_t koff = (int64_t)t * 128;
        if (CONV && (kind == 0 || kind == 3) && p.cUni) {
            // C % 64 == 0: one tap per K-tile - scalar tap / offset, a lane only tests its mask bit (see gemm_nt_128)
            const int kq0 = t * 64;
            const int tap = (int)__umulhi((unsigned)kq0, p.cMagic);
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
            const int64_t soff = ((int64_t)(ky * p.cW + kx) * p.cC + (kq0 - tap * p.cC)) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = tap < 9 && ((vmask[kind == 3][j] >> tap) & 1u);
                const char *sp = ok ? src[kind][j] + soff : (const char *)hgr_zero_page;
                __builtin_amdgcn_global_load_lds((const AS1 void *)sp, (AS3 void *)(dst + j * 8192), 16, 0, 0);
            }
            return;
        }
        if (CONV && (kind == 0 || kind == 3)) {
            // implicit im2col (see gemm_nt_128): K index = tap * C + channel, 8 channels per 16-B chunk, OOB taps -> zero page
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kq = t * 64 + cchunk[j] * 8;
                const int tap = (int)__umulhi((unsigned)kq, p.cMagic), cin = kq - tap * p.cC;
                const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
                const bool ok = tap < 9 && ((vmask[kind == 3][j] >> tap) & 1u);
                const char *sp = ok ? src[kind][j] + ((int64_t)(ky * p.cW + kx) * p.cC + cin) * 2 : (const char *)hgr_zero_page;
                __builtin_amdgcn_global_load_lds((const AS1 void *)sp, (AS3 void *)(dst + j * 8192), 16, 0, 0);
            }
            return;
        }
        __builtin_amdgcn_global_load_lds((const AS1 void *)(src[kind][0] + koff), (AS3 void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const AS1 void *)(src[kind][1] + koff), (AS3 void *)(dst + 8192), 16, 0, 0);
    };

    f32x4 acc[2][2][4][2];      // [m-half][n-half][m tile][n tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;    // >= 2 (host guarantees)
    // prologue: K-tile 0 complete, K-tile 1 without A1 (sequence numbers 0..6)
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    issue(0, 1); issue(1, 1); issue(2, 1);
    HGR_RWAIT(10);              // A0(0), W0(0) landed
    if (wm) HGR_MBAR();         // ping-pong: group 1 runs one barrier interval behind group 0

    const int offA = (wm * 64 + r) * 128;      // + m tile * 2048, within pieces A0 / A1
    const int offW = (wn * 32 + r) * 128;      // + n tile * 2048, within pieces W0 / W1
    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;

    vec8 af[4][2], wf0[2][2], wf1[2][2];

    // MODE 0: steady state, 1: second-last K-tile (nothing left to issue after ph1), 2: last K-tile.
    // Every phase = R interval (reads, 1 piece issued, wait) | barrier | M interval (16 MFMAs) | barrier.
    auto ktile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char *buf = smem + (t & 1) * (4 * PIECE);
        // ---- ph1: Q(0,0) ----------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf0[j][0] = *(const vec8 *)(buf + 1 * PIECE + offW + j * 2048 + sw0);
            wf0[j][1] = *(const vec8 *)(buf + 1 * PIECE + offW + j * 2048 + sw1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const vec8 *)(buf + 0 * PIECE + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(buf + 0 * PIECE + offA + i * 2048 + sw1);
        }
        if (MODE <= 1) issue(3, t + 1);
        if (MODE <= 1) HGR_RWAIT(10); else HGR_RWAIT(2);          // W1(t) landed
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[0][0][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph2: Q(0,1) ----------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf1[j][0] = *(const vec8 *)(buf + 2 * PIECE + offW + j * 2048 + sw0);
            wf1[j][1] = *(const vec8 *)(buf + 2 * PIECE + offW + j * 2048 + sw1);
        }
        if (MODE == 0) issue(0, t + 2);
        if (MODE == 0) HGR_RWAIT(10); else if (MODE == 1) HGR_RWAIT(8); else HGR_RWAIT(0);   // A1(t) landed
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[0][1][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph3: Q(1,1) ----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const vec8 *)(buf + 3 * PIECE + offA + i * 2048 + sw0);
            af[i][1] = *(const vec8 *)(buf + 3 * PIECE + offA + i * 2048 + sw1);
        }
        if (MODE == 0) issue(1, t + 2);
        HGR_RBAR();                                               // ph4 reads nothing new
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][1][i][j] = T16<DT>::mfma16(wf1[j][kk], af[i][kk], acc[1][1][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
        // ---- ph4: Q(1,0) ----------------------------------------------------------------------
        if (MODE == 0) issue(2, t + 2);
        if (MODE == 0) HGR_RWAIT(10); else if (MODE == 1) HGR_RWAIT(4); else HGR_RBAR();     // A0(t+1), W0(t+1) landed
        __builtin_amdgcn_s_setprio(1);
        if (do_mma)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][0][i][j] = T16<DT>::mfma16(wf0[j][kk], af[i][kk], acc[1][0][i][j]);
        __builtin_amdgcn_s_setprio(0);
        HGR_MBAR();
    };

    for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>());
    ktile(nk - 2, std::integral_constant<int, 1>());
    ktile(nk - 1, std::integral_constant<int, 2>());
    if (!wm) HGR_MBAR();
    if (HGR_LAB_ON(p.dbg == 3)) { if (acc[0][0][0][0][0] == 123.456f) ((float *)p.C)[0] = 1.f; return; }
    // group 0 waits for group 1's last interval: every LDS read is done, no DMA in flight

    // tile (a, b, i, j) of this lane holds C[m][n .. n+3],
    //   m = m0 + wm*128 + a*64 + i*16 + r,   n = n0 + wn*64 + b*32 + j*16 + g*4
    if (!OUT32 && p.vec_ok && (p.ldc & 7) == 0 && (EPI <= HGR_EPI_BIAS_QUICKGELU || EPI == HGR_EPI_BIAS_RELU)) {
        // 16-bit output: transpose the wave's 128 x 64 tile through its private LDS slice (rows of 128 B
        // + 16 B pad) and write full 128-byte lines with 16-byte stores (8 lanes per row, 8 rows per
        // instruction) instead of 32-byte fragments of 16 different lines per instruction.
        constexpr int RS = 144;
        char *my = smem + wave * (128 * RS);
        // FULL = the whole 256 x 256 tile is inside C: no guards, so the 4 bias loads and the 16 stores are independent
        // (guarded loads make hipcc wait vmcnt(0) behind every one of them: 4 serial L2 round trips per tile)
        auto staged = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            f32x4 bq[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + b * 32 + j * 16 + g * 4;
                bq[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (epi_has_bias(EPI)) {
                    if (FULL || n + 3 < p.N) bq[b][j] = *(const f32x4 *)(p.bias + n);
                    else
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) bq[b][j][e] = p.bias[n + e];
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
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
            // LDS ops of one wave complete in order, and the slice is private to the wave: no barrier needed
            const int ch = lane & 7, rr = lane >> 3;
            const int nb = n0 + wn * 64 + ch * 8;
            E *dst0 = (E *)p.C + (int64_t)(m0 + wm * 128 + rr) * p.ldc + nb;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = q * 8 + rr;
                const int m = m0 + wm * 128 + row;
                const u32x4 v = *(const u32x4 *)(my + row * RS + ch * 16);
                if (HGR_LAB_ON(p.dbg == 4)) { if (v[0] == 0x12345678u) ((float *)p.C)[0] = 1.f; continue; }
                E *dst = dst0 + (int64_t)q * 8 * p.ldc;
                if (FULL) *(u32x4 *)dst = v;
                else if (m < p.M) {
                    if (nb + 7 < p.N) *(u32x4 *)dst = v;
                    else {
                        const E *ve = (const E *)&v;
                        for (int e = 0; e < 8 && nb + e < p.N; ++e) dst[e] = ve[e];
                    }
                }
            }
        };
        if (m0 + 256 <= p.M && n0 + 256 <= p.N) staged(std::true_type());
        else staged(std::false_type());
        return;
    }
    if (p.vec_ok && m0 + 256 <= p.M && n0 + 256 <= p.N) {
        f32x4 bq[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bq[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (epi_has_bias(EPI)) bq[b][j] = *(const f32x4 *)(p.bias + n0 + wn * 64 + b * 32 + j * 16 + g * 4);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {       // batches of 8 quads: all addend loads first, then the stores
            f32x4 ad[2][2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ad[i][b][j] = load_addend<DT, EPI>(p, m0 + wm * 128 + a * 64 + (ih * 2 + i) * 16 + r, n0 + wn * 64 + b * 32 + j * 16 + g * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                store_quad_full<DT, EPI, OUT32>(p, acc[a][b][ih * 2 + i][j], bq[b][j], ad[i][b][j],
                                                m0 + wm * 128 + a * 64 + (ih * 2 + i) * 16 + r, n0 + wn * 64 + b * 32 + j * 16 + g * 4);
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + a * 64 + i * 16 + r;
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


namespace {
template <int DT, int EPI>
void plain256(const GemmArgs &a, bool out32, dim3 grid, hipStream_t s) {
    if constexpr (EPI == HGR_EPI_QGELU_GRAD16 || EPI == HGR_EPI_BIAS_ADD16_RELU) {
        hipLaunchKernelGGL((gemm_nt_256<DT, EPI, false>), grid, dim3(NT256), 0, s, a);
    } else {
        if (out32) hipLaunchKernelGGL((gemm_nt_256<DT, EPI, true>), grid, dim3(NT256), 0, s, a);
        else hipLaunchKernelGGL((gemm_nt_256<DT, EPI, false>), grid, dim3(NT256), 0, s, a);
    }
}
template <int DT>
void launch_256_dt(const GemmArgs &a, int epi, bool out32, bool conv, dim3 grid, hipStream_t s) {
    if (conv) { hipLaunchKernelGGL((gemm_nt_256<DT, HGR_EPI_BIAS_RELU, false, true>), grid, dim3(NT256), 0, s, a); return; }
    switch (epi) {
        case HGR_EPI_NONE: plain256<DT, HGR_EPI_NONE>(a, out32, grid, s); break;
        case HGR_EPI_BIAS: plain256<DT, HGR_EPI_BIAS>(a, out32, grid, s); break;
        case HGR_EPI_BIAS_QUICKGELU: plain256<DT, HGR_EPI_BIAS_QUICKGELU>(a, out32, grid, s); break;
        case HGR_EPI_BIAS_RELU: plain256<DT, HGR_EPI_BIAS_RELU>(a, out32, grid, s); break;
        case HGR_EPI_BIAS_ADD16_RELU: plain256<DT, HGR_EPI_BIAS_ADD16_RELU>(a, out32, grid, s); break;
        case HGR_EPI_ACCUM: plain256<DT, HGR_EPI_ACCUM>(a, out32, grid, s); break;
        case HGR_EPI_QGELU_GRAD16: plain256<DT, HGR_EPI_QGELU_GRAD16>(a, out32, grid, s); break;
        default: plain256<DT, HGR_EPI_BIAS_RESIDUAL>(a, out32, grid, s); break;
    }
}
}  // namespace

void launch_256(const GemmArgs &a, int dtype, int epi, bool out32, bool conv, dim3 grid, hipStream_t s) {
    if (dtype == HGR_BF16) launch_256_dt<HGR_BF16>(a, epi, out32, conv, grid, s);
    else launch_256_dt<HGR_F16>(a, epi, out32, conv, grid, s);
}

}  // namespace hgr_gemm
