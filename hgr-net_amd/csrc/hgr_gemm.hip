// hgr_gemm_nt: C[M,N] = epilogue(A[M,K] . W[N,K]^T) on the gfx950 matrix cores.
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out, in]), so a lane's
// MFMA fragment (8 consecutive k of one row) is one 16-byte LDS read for either operand.
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2(M) x 2(N), each wave 64 x 64 =
// 4 x 4 tiles of v_mfma_f32_16x16x32.  W is fed as the MFMA "A" operand and the activations as
// "B", i.e. the wave computes C^T tiles: the 4 accumulator registers of a lane are then 4
// CONSECUTIVE n of one output row m, so bias/residual/stores are 8- or 16-byte vector accesses.
//
// Staging: global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) into two LDS stages of 32 KB; the LDS
// image is lane-linear ([row][8 chunks of 16 B], 128-B rows), the bank-conflict swizzle
// chunk' = chunk ^ (row & 7) is applied on the per-lane SOURCE address and again on the ds_read
// address (cdna_hip_programming.md rule 21).  With it every ds_read_b128 of a fragment is
// conflict-free: a 16-lane read group touches 16 distinct 16-B slots of the 256-B bank row.
// Loop: issue stage t+1, compute stage t, vmcnt(0) + one barrier per K-tile ("minimum 2-phase").
//
// Block -> tile map is XCD-aware: consecutive tile ids (which share an operand panel) are dealt to
// ONE XCD (blocks b and b+8 share an XCD's L2), and the fastest-varying tile index is chosen on the
// host so that the larger operand is read from HBM once (see hgr_gemm_nt below).
#include "hgr_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;       // one operand, one stage: 16 KB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + W
constexpr int NT = 256;

struct GemmArgs {
    const char *A; int64_t lda;
    const char *W; int64_t ldw;
    void *C; int64_t ldc;
    const float *bias;
    const float *res; int64_t ldr;
    int M, N, K;
    int tiles_m, tiles_n;
    int m_fastest;   // 1: consecutive tile ids walk M first (W panel shared), 0: walk N first
    int vec_ok;      // C / residual rows allow 4-element vector access
};

__device__ __forceinline__ float quick_gelu(float v) { return v / (1.0f + __expf(-1.702f * v)); }

template <int DT, int EPI, bool OUT32>
__global__ __launch_bounds__(NT) void gemm_nt_128(GemmArgs p) {
    typedef typename T16<DT>::vec8 vec8;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, g = lane >> 4;

    // XCD-aware, bijective remap of the 1-D grid (cdna_hip_programming.md T1)
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    int tm, tn;
    if (p.m_fastest) { tn = wg / p.tiles_m; tm = wg - tn * p.tiles_m; }
    else             { tm = wg / p.tiles_n; tn = wg - tm * p.tiles_n; }
    const int m0 = tm * BM, n0 = tn * BN;

    // per-lane source rows of the 4 + 4 LDS-DMA pieces this thread issues per stage
    const char *srcA[4], *srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = (i * 4 + wave) * 64 + lane;    // 16-B chunk id inside the 128 x 64 tile
        const int row = id >> 3, c = (id & 7) ^ (row & 7);
        const int gm = min(m0 + row, p.M - 1);        // edge rows: load a valid row, never store it
        const int gn = min(n0 + row, p.N - 1);
        srcA[i] = p.A + ((int64_t)gm * p.lda + c * 8) * 2;
        srcW[i] = p.W + ((int64_t)gn * p.ldw + c * 8) * 2;
    }

    auto stage = [&](int buf, int kt) {
        char *sA = smem + buf * STAGE_BYTES;
        char *sW = sA + TILE_BYTES;
        const int64_t koff = (int64_t)kt * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void *)(srcA[i] + koff),
                                             (AS3 void *)(sA + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void *)(srcW[i] + koff),
                                             (AS3 void *)(sW + (i * 4 + wave) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment read offsets: row & 7 == r & 7 for every fragment of this lane
    const int offW = (wn * 64 + r) * 128;
    const int offA = (wm * 64 + r) * 128;
    const int sw0 = ((0 + g) ^ (r & 7)) * 16;
    const int sw1 = ((4 + g) ^ (r & 7)) * 16;

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char *sA = smem + cur * STAGE_BYTES;
        const char *sW = sA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = kk ? sw1 : sw0;
            vec8 wf[4], xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wf[t] = *(const vec8 *)(sW + offW + t * 2048 + sw);
                xf[t] = *(const vec8 *)(sA + offA + t * 2048 + sw);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T16<DT>::mfma16(wf[i], xf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane holds, for tile (i, j), C[m][n .. n+3] with
    //   m = m0 + wm*64 + j*16 + r,   n = n0 + wn*64 + i*16 + g*4
    typedef typename T16<DT>::elem E;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + g * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            const bool full = (n + 3 < p.N);
            if (EPI != HGR_EPI_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (full || n + e < p.N) v[e] += p.bias[n + e];
            }
            if (EPI == HGR_EPI_BIAS_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
            }
            if (full && p.vec_ok) {
                if (EPI == HGR_EPI_BIAS_RESIDUAL) {
                    const f32x4 rr = *(const f32x4 *)(p.res + (int64_t)m * p.ldr + n);
                    v += rr;
                }
                if (OUT32) {
                    *(f32x4 *)((float *)p.C + (int64_t)m * p.ldc + n) = v;
                } else {
                    *(typename T16<DT>::vec4 *)((E *)p.C + (int64_t)m * p.ldc + n) = cvt4<DT>(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) break;
                    float x = v[e];
                    if (EPI == HGR_EPI_BIAS_RESIDUAL) x += p.res[(int64_t)m * p.ldr + n + e];
                    if (OUT32) ((float *)p.C)[(int64_t)m * p.ldc + n + e] = x;
                    else ((E *)p.C)[(int64_t)m * p.ldc + n + e] = (E)x;
                }
            }
        }
    }
}

template <int DT, int EPI>
void launch_epi(const GemmArgs &a, bool out32, dim3 grid, hipStream_t s) {
    if (out32) hipLaunchKernelGGL((gemm_nt_128<DT, EPI, true>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((gemm_nt_128<DT, EPI, false>), grid, dim3(NT), 0, s, a);
}

template <int DT>
void launch_dt(const GemmArgs &a, int epi, bool out32, dim3 grid, hipStream_t s) {
    switch (epi) {
        case HGR_EPI_NONE: launch_epi<DT, HGR_EPI_NONE>(a, out32, grid, s); break;
        case HGR_EPI_BIAS: launch_epi<DT, HGR_EPI_BIAS>(a, out32, grid, s); break;
        case HGR_EPI_BIAS_QUICKGELU: launch_epi<DT, HGR_EPI_BIAS_QUICKGELU>(a, out32, grid, s); break;
        default: launch_epi<DT, HGR_EPI_BIAS_RESIDUAL>(a, out32, grid, s); break;
    }
}

}  // namespace

extern "C" int hgr_gemm_nt(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                           const float *bias, const float *residual, int64_t ldr,
                           int M, int N, int K, int dtype, int epilogue, int out_f32, void *stream) {
    HGR_REQUIRE(A && W && C, "hgr_gemm_nt: null operand");
    HGR_REQUIRE(M >= 1 && N >= 1 && K >= BK, "hgr_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    HGR_REQUIRE(K % BK == 0, "hgr_gemm_nt: K=%d must be a multiple of %d (pad the operands)", K, BK);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "hgr_gemm_nt: lda=%lld ldw=%lld must be >= K and multiples of 8", (long long)lda, (long long)ldw);
    HGR_REQUIRE(hgr_aligned(A, 16) && hgr_aligned(W, 16), "hgr_gemm_nt: A and W must be 16-byte aligned");
    HGR_REQUIRE(ldc >= N, "hgr_gemm_nt: ldc=%lld < N=%d", (long long)ldc, N);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_gemm_nt: bad dtype %d", dtype);
    HGR_REQUIRE(epilogue >= HGR_EPI_NONE && epilogue <= HGR_EPI_BIAS_RESIDUAL, "hgr_gemm_nt: bad epilogue %d", epilogue);
    HGR_REQUIRE(epilogue == HGR_EPI_NONE || bias, "hgr_gemm_nt: epilogue %d needs bias", epilogue);
    HGR_REQUIRE(epilogue != HGR_EPI_BIAS_RESIDUAL || (residual && ldr >= N), "hgr_gemm_nt: residual epilogue needs residual with ldr >= N");
    HGR_REQUIRE(hgr_aligned(C, out_f32 ? 4 : 2), "hgr_gemm_nt: C misaligned");

    GemmArgs a;
    a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw;
    a.C = C; a.ldc = ldc; a.bias = bias; a.res = residual; a.ldr = ldr;
    a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM;
    a.tiles_n = (N + BN - 1) / BN;
    // each XCD owns a contiguous range of tile ids; the operand indexed by the slow tile index is
    // fetched ~once, the other one once per XCD.  Make the bigger operand the once-fetched one.
    a.m_fastest = ((int64_t)N * K > (int64_t)M * K) ? 1 : 0;
    bool vec = (ldc % 4 == 0) && hgr_aligned(C, out_f32 ? 16 : 8);
    if (epilogue == HGR_EPI_BIAS_RESIDUAL) vec = vec && (ldr % 4 == 0) && hgr_aligned(residual, 16);
    a.vec_ok = vec ? 1 : 0;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == HGR_BF16) launch_dt<HGR_BF16>(a, epilogue, out_f32 != 0, grid, s);
    else launch_dt<HGR_F16>(a, epilogue, out_f32 != 0, grid, s);
    HGR_CHECK_LAUNCH("hgr_gemm_nt");
    return HGR_OK;
}
