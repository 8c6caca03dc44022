// Host entry points of the NT GEMM family: argument validation, tile-plan choice, launch (kernels: hgr_gemm_128 / _256 / _duo .hip).
#include "hgr_gemm_common.h"

using namespace hgr_gemm;

int hgr_conv3x3_c32_launch(const void *x, const void *w, const float *bias, void *out, int B, int H, int W, int Cout, int Kp,
                           int dtype, int relu, void *stream, int pool, int C);      // hgr_conv_direct.hip

namespace {
constexpr int BM = 128, BN = 128;

// development knobs of gemm_nt_duo, read once: HGR_GEMM_GROUP (raster group; default by shape, duo_group_for), HGR_GEMM_DBG (bit 8: sc1 output stores)
int duo_group_env() {
    static int g = -1;
    if (g < 0) { const char *e = getenv("HGR_GEMM_GROUP"); g = e ? atoi(e) : 0; if (g < 0) g = 0; }
    return g;
}
int duo_group() { return duo_group_env() ? duo_group_env() : 4; }          // gemm_nt_ws (no plan)
// Row panels per raster group of a gemm_nt_duo launch (duo_apply_plan).  By shape unless HGR_GEMM_GROUP=n forces n: 4 (64 tiles in flight per
// XCD = 4 row panels x 16 column panels), but ONE for launches of at most 8 column panels - the few tiles that share an activation
// panel are then dispatched back to back instead of 4 slots apart.  Measured (tools/raster_split_ab.sh, FETCH_SIZE per launch, ViT-B/32
// step): the N = 768 producers 250 -> 222 MB (group 2: 232, 8: 280), patch GEMM 345 -> < 300; c_fc (24 column panels) 207 MB at 4,
// 241 / 270 at 2 / 1, 219 at 8.  Step times of all three BASELINE configurations unchanged (tools/group_ab.sh: 4.953 vs 4.949 ms,
// 10.08 vs 10.09 ms, 224.1 vs 224.8 ms): this trims fabric traffic, not time.
int duo_group_for(int tiles_n) { return duo_group_env() ? duo_group_env() : (tiles_n <= 8 ? 1 : 4); }
int duo_dbg() {
    static int d = -1;
    if (d < 0) d = hgr_lab_env("HGR_GEMM_DBG");          // lab builds only (hgr_common.h): 0 in libhgr.so
    return d;
}

// Tail plan of gemm_nt_duo.  The chip runs 512 of its workgroups at a time (2 per CU); a launch of T > 512 full tiles that is not
// a whole number of rounds ends in a partly empty round: the N = 768 residual producers of ViT-B/32 at batch 512 (600 tiles) spent
// 25 - 29 % of their time with 88 workgroups on the chip (tools/round_staircase.py).  The plan gives the LAST row panels to half
// tiles (128 x 128, dispatched after the full tiles): work of half the size fills the slots the full tiles free first.  Chosen by
// list-scheduling the two tile kinds on 512 slots (a half tile priced at HGR_DUO_HALF_COST = 0.55 of a full one: half the MFMAs, the
// same per-tile overhead) over the candidate panel counts; HGR_DUO_TAIL=0 switches it off, HGR_DUO_PB=n forces n full panels.
struct DuoPlan { int big_panels, nbig, tiles_m_half, grid; };
int tail_env = -2, pb_env = -2;              // hgr_gemm_set_tail / HGR_DUO_TAIL, HGR_DUO_PB
double half_cost = 0.55;
DuoPlan duo_plan(int M, int N, bool allow_tail) {
    if (tail_env == -2) {
        const char *e = getenv("HGR_DUO_TAIL"); tail_env = e ? atoi(e) : 1;
        const char *f = getenv("HGR_DUO_PB"); pb_env = f ? atoi(f) : -1;
        const char *h = getenv("HGR_DUO_HALF_COST"); if (h && atof(h) > 0.1) half_cost = atof(h);
    }
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 127) / 128, S = 512;
    const int64_t T = (int64_t)tiles_m * tiles_n;
    DuoPlan best{tiles_m, (int)T, 0, (int)T};
    if (allow_tail && tail_env && pb_env < 0 && M > 128 && (T <= S / 4 || (tiles_m <= 2 && T <= S))) {
        // a launch that covers a fraction of the chip (the class-token GEMMs of a ViT's last block: 512 rows) is bound by what ONE
        // workgroup can pull per K-tile, not by the matrix cores: all half tiles = twice the workgroups, 2/3 of the bytes per K-tile each.
        // Likewise at most two row panels on fewer tiles than slots (the class-logits GEMM and hgr_logits_eval at batch 512: 356
        // full tiles on 256 CUs -> 712 half tiles; measured 48.0 -> 46.2 us fused, 21.6 -> 21.0 us plain, bit-identical)
        best.big_panels = 0; best.nbig = 0; best.tiles_m_half = (M + 127) / 128; best.grid = best.tiles_m_half * tiles_n;
        return best;
    }
    if (!allow_tail || !tail_env || T <= S || T % S == 0) return best;
    // measured (tools/tail_sweep.py, ViT-B/32 tower launches at batch 512, cold operands): last round 15 - 17 % full (out, proj, patch:
    // 600 / 588 tiles) -13 %, -14.5 %, -11 %; last round 52 % full (qkv, 1800 tiles) +-0; 69 % full (fc, 2400 tiles) +2 %: a
    // half-full last round already overlaps well, and half tiles stage fewer flops per byte
    if (pb_env < 0 && (T % S) * 100 > 45 * (int64_t)S) return best;
    auto makespan = [&](int pb) {
        const int64_t nb = (int64_t)pb * tiles_n;
        const int mh = M - pb * 256;
        int64_t nh = mh > 0 ? (int64_t)((mh + 127) / 128) * tiles_n : 0;
        const int64_t q = nb / S, rem = nb % S;
        double tE = (double)q, tL = (double)q + 1.0;         // S - rem slots are free at q, rem slots at q + 1
        int64_t nE = S - rem, nL = rem;
        double end = rem ? tL : tE;
        if (!nb) end = 0.0;
        while (nh > 0) {
            const bool useE = nE > 0 && (nL == 0 || tE <= tL);
            double &t = useE ? tE : tL;
            const int64_t k = useE ? nE : nL;
            t += half_cost;
            if (t > end) end = t;
            nh -= k;
        }
        return end;
    };
    int pick = tiles_m;
    double bm = makespan(tiles_m);
    if (pb_env >= 0) pick = pb_env < tiles_m ? pb_env : tiles_m;
    else
        for (int pb = tiles_m - 1; pb >= 0 && pb >= tiles_m - 96; --pb) {
            const double m = makespan(pb);
            // (forced panel counts, tools/producer_ab.py with warm clocks and interleaved arms, out_proj K = 768 / c_proj K = 3072 of
            // ViT-B/32 at batch 512, us: no tail 61.7 / 150.0, 85 full panels - this loop's pick - 55.7 / 132.6, 70: 57.6 / 141.4,
            // 55: 57.4 / 142.5: more half tiles than the last round needs cost time even on the epilogue-bound K = 768 launch)
            if (m < bm - 1e-9) { bm = m; pick = pb; }
        }
    if (pick == tiles_m) return best;
    const int mh = M - pick * 256;
    best.big_panels = pick; best.nbig = pick * tiles_n; best.tiles_m_half = (mh + 127) / 128;
    best.grid = best.nbig + best.tiles_m_half * tiles_n;
    return best;
}
void duo_apply_plan(GemmArgs &a, bool allow_tail, dim3 &grid) {
    const DuoPlan pl = duo_plan(a.M, a.N, allow_tail);
    a.nbig = pl.nbig; a.big_panels = pl.big_panels; a.tiles_m_half = pl.tiles_m_half;
    a.group = a.m_fastest ? duo_group() : duo_group_for((a.N + 127) / 128);
    if (duo_dbg() & 64) fprintf(stderr, "[duo_plan] M=%d N=%d K=%d -> %d full panels (%d tiles) + %d half panels, grid %d\n",
                                     a.M, a.N, a.K, pl.big_panels, pl.nbig, pl.tiles_m_half, pl.grid);
    grid = dim3((unsigned)pl.grid);
}

// tile plan override (hgr_gemm_set_tile); HGR_GEMM_TILE=128|256|2 sets the initial value
int g_force_tile = -1;
int hgr_gemm_force_tile() {
    if (g_force_tile < 0) { const char *e = getenv("HGR_GEMM_TILE"); g_force_tile = e ? atoi(e) : 0; }
    return g_force_tile;
}

}  // namespace

extern "C" int hgr_gemm_set_tail(int enabled, int full_panels) {
    HGR_REQUIRE((enabled == 0 || enabled == 1) && full_panels >= -1, "hgr_gemm_set_tail: enabled must be 0 or 1, full_panels >= -1 (-1 = choose), got %d, %d", enabled, full_panels);
    duo_plan(1, 1, false);                       // reads the environment once, so that it cannot overwrite this call later
    const int prev = tail_env;
    tail_env = enabled; pb_env = full_panels;
    return prev;
}

extern "C" int hgr_gemm_set_tile(int tile) {
    HGR_REQUIRE(tile == 0 || tile == 128 || tile == 256 || tile == 2, "hgr_gemm_set_tile: tile must be 0, 128, 256 or 2 (256 x 128 tiles, two workgroups per CU), got %d", tile);
    const int prev = hgr_gemm_force_tile();
    g_force_tile = tile;
    return prev;
}

extern "C" int hgr_gemm_set_ws(int enabled) {
    HGR_REQUIRE(enabled == 0 || enabled == 1, "hgr_gemm_set_ws: enabled must be 0 or 1, got %d", enabled);
    return ws_set(enabled);
}

extern "C" int hgr_gemm_set_p8(int mode) {
    HGR_REQUIRE(mode >= 0 && mode <= 2, "hgr_gemm_set_p8: mode must be 0 (never), 1 (wherever it covers) or 2 (by shape), got %d", mode);
    return p8_set(mode);
}

extern "C" int hgr_gemm_set_persist(int enabled) {
    HGR_REQUIRE(enabled == 0 || enabled == 1, "hgr_gemm_set_persist: enabled must be 0 or 1, got %d", enabled);
    return duo_set_persist(enabled);
}

extern "C" int hgr_gemm_nt(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                           const float *bias, const void *residual, int64_t ldr,
                           int M, int N, int K, int dtype, int epilogue, int out_f32, void *stream) {
    HGR_REQUIRE(A && W && C, "hgr_gemm_nt: null operand");
    HGR_REQUIRE(M >= 1 && N >= 1 && K >= BK, "hgr_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    HGR_REQUIRE(K % BK == 0, "hgr_gemm_nt: K=%d must be a multiple of %d (pad the operands)", K, BK);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "hgr_gemm_nt: lda=%lld ldw=%lld must be >= K and multiples of 8", (long long)lda, (long long)ldw);
    HGR_REQUIRE(hgr_aligned(A, 16) && hgr_aligned(W, 16), "hgr_gemm_nt: A and W must be 16-byte aligned");
    HGR_REQUIRE(ldc >= N, "hgr_gemm_nt: ldc=%lld < N=%d", (long long)ldc, N);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_gemm_nt: bad dtype %d", dtype);
    HGR_REQUIRE(epilogue >= HGR_EPI_NONE && epilogue <= HGR_EPI_QGELU_GRAD16, "hgr_gemm_nt: bad epilogue %d", epilogue);
    HGR_REQUIRE(epilogue != HGR_EPI_ACCUM || out_f32, "hgr_gemm_nt: ACCUM accumulates into an fp32 C");
    HGR_REQUIRE(!epi_has_bias(epilogue) || bias, "hgr_gemm_nt: epilogue %d needs bias", epilogue);
    HGR_REQUIRE((epilogue != HGR_EPI_BIAS_RESIDUAL && !epi_has_idn16(epilogue)) || (residual && ldr >= N), "hgr_gemm_nt: epilogue %d needs its second operand (residual / identity / pre-activation) with ldr >= N", epilogue);
    HGR_REQUIRE(!epi_has_idn16(epilogue) || !out_f32, "hgr_gemm_nt: ADD16_RELU / QGELU_GRAD16 write 16-bit output");
    HGR_REQUIRE(hgr_aligned(C, out_f32 ? 4 : 2), "hgr_gemm_nt: C misaligned");

    bool vec = (ldc % 4 == 0) && hgr_aligned(C, out_f32 ? 16 : 8);
    if (epi_has_bias(epilogue)) vec = vec && hgr_aligned(bias, 16);
    if (epilogue == HGR_EPI_BIAS_RESIDUAL) vec = vec && (ldr % 4 == 0) && hgr_aligned(residual, 16);
    if (epi_has_idn16(epilogue)) vec = vec && (ldr % 4 == 0) && hgr_aligned(residual, 8);
    static int dbg = -1, split_env = -1;
    if (dbg < 0) dbg = hgr_lab_env("HGR_GEMM_DBG");
    if (split_env < 0) { const char *e = getenv("HGR_GEMM_SPLIT"); split_env = e ? atoi(e) : 1; }
    hipStream_t s = (hipStream_t)stream;
    const size_t csz = out_f32 ? 4 : 2, rsz = epi_has_idn16(epilogue) ? 2 : 4;

    // launch rows [m_lo, m_lo + m_cnt) with one tile size
    auto launch = [&](int m_lo, int m_cnt, bool big) {
        GemmArgs a;
        a.A = (const char *)A + (size_t)m_lo * lda * 2; a.lda = lda; a.W = (const char *)W; a.ldw = ldw;
        a.C = (char *)C + (size_t)m_lo * ldc * csz; a.ldc = ldc; a.bias = bias;
        a.res = residual ? (const float *)((const char *)residual + (size_t)m_lo * ldr * rsz) : nullptr; a.ldr = ldr;
        a.M = m_cnt; a.N = N; a.K = K;
        const int T = big ? 256 : 128;
        a.tiles_m = (m_cnt + T - 1) / T;
        a.tiles_n = (N + T - 1) / T;
        // each XCD owns a contiguous range of tile ids; the operand indexed by the slow tile index is fetched ~once,
        // the other one once per XCD.  Make the bigger operand the once-fetched one.
        a.m_fastest = ((int64_t)N * K > (int64_t)m_cnt * K) ? 1 : 0;
        a.vec_ok = vec ? 1 : 0;
        a.dbg = dbg; a.kc = 0; a.csplit = 0;
        // outputs at most 64 wide (1x1 convolutions into the 64-channel ResNet stages): the 256 x 64 arrangement of the small kernel
        if (!big && N <= 64 && m_cnt >= 1024 && epilogue == HGR_EPI_BIAS_RELU && !out_f32) {
            a.tiles_m = (m_cnt + 255) / 256; a.tiles_n = 1;
            launch_128(a, dtype, HGR_EPI_BIAS_RELU, false, V128_TALL, dim3((unsigned)a.tiles_m), s);
            return;
        }
        dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
        if (big) launch_256(a, dtype, epilogue, out_f32 != 0, false, grid, s);
        else launch_128(a, dtype, epilogue, out_f32 != 0, V128_PLAIN, grid, s);
    };

    // Tile choice.  The 256^2 deep-pipelined kernel owns a CU (one 512-thread workgroup), so a launch runs in rounds of
    // 256 workgroups; 128^2 tiles run 2 workgroups per CU.  Three plans are priced with measured tile times: all big,
    // all small, or full rounds of big tiles on the first row panels + the remaining panels on the small-tile kernel in a
    // second launch (pays off at long K, where a mostly empty last big round is expensive: c_proj / patch GEMM).
    const int force = hgr_gemm_force_tile();
    const int tn256 = (N + 255) / 256, tm256 = (M + 255) / 256;
    const int64_t t256 = (int64_t)tm256 * tn256;
    const int64_t t128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    // measured tile times on MI355X (us): 256^2 tile ~ 1.75 per K-tile + 14 (prologue + epilogue, nothing overlaps them
    // at one workgroup per CU); 128^2 tile at 2 per CU ~ 1.08 per K-tile + 9
    const double Tb = 1.75 * (K / 64) + 14.0, Ts = 1.08 * (K / 64) + 9.0;
    const double cost_small = (double)((t128 + 511) / 512) * Ts;
    const double cost_big = (double)((t256 + 255) / 256) * Tb;
    const int64_t rounds = t256 / 256;
    const int big_panels = (int)((rounds * 256) / tn256);
    const int m1 = big_panels * 256;
    double cost_split = 1e30;
    if (split_env && rounds >= 1 && m1 > 0 && m1 < M) {
        const int64_t ts = (int64_t)((M - m1 + 127) / 128) * ((N + 127) / 128);
        cost_split = (double)rounds * Tb + (double)((ts + 511) / 512) * Ts + 2.0;          // + one kernel boundary
    }
    // 256 x 128 tiles, two workgroups per CU (gemm_nt_duo): fp32 residual / 16-bit epilogues of the transformer towers
    const bool duo_ok = K >= 128 && (out_f32 || (epilogue != HGR_EPI_BIAS_RESIDUAL && epilogue != HGR_EPI_ACCUM)) &&
                        (int64_t)M * lda * 2 < (1ll << 32) && (int64_t)N * ldw * 2 < (1ll << 32) && ldc < (1 << 20) && ldr < (1 << 20);
    auto launch_d = [&]() {
        GemmArgs a;
        a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias;
        a.res = (const float *)residual; a.ldr = ldr; a.M = M; a.N = N; a.K = K;
        a.tiles_m = (M + 255) / 256; a.tiles_n = (N + 127) / 128;
        a.m_fastest = ((int64_t)N * K > (int64_t)M * K) ? 1 : 0;
        a.vec_ok = vec ? 1 : 0; a.dbg = dbg; a.kc = 0; a.csplit = 0; a.group = duo_group();
        dim3 grid;
        duo_apply_plan(a, true, grid);
        launch_duo(a, dtype, epilogue, out_f32 != 0, 0, grid, s);
    };
    // Plan choice for the shapes gemm_nt_duo covers: measured (tools/gemm_plan_ab.py, same-process A/B, f16, one MI355X; bit-identical
    // outputs): qkv 97 -> 91 us, out-proj 59 -> 49, c_fc 136 -> 124, c_proj 145 -> 126, patch 124 -> 118, class logits 24.7 -> 20.7,
    // text qkv 168 -> 154, ViT-L/14 out / proj 411 -> 361 / 1029 -> 999, ViT-L/14 qkv / fc a tie; a launch with fewer tiles than
    // half the chip's 512 slots (small text batches, ragged test shapes) stays on the 128^2 / cost-model plans.
    const int64_t tduo = (int64_t)((M + 255) / 256) * ((N + 127) / 128);
    // ... and an output much narrower than its 128-column tiles (the 64-channel ResNet stage: half of every tile would be padding;
    // measured 252 vs 194 us on the 256 x 64 arrangement of the small kernel) stays where it was
    const bool duo_fits = (N + 127) / 128 * 128 - N <= N / 8;
    // the role-split kernel (gemm_nt_ws: the epilogue runs in helper waves under the next tile's MFMAs) for 16-bit outputs made of whole tiles
    // (not BIAS_QUICKGELU: in gemm_nt_duo's plain f16 instantiation hipcc fuses the last product of QuickGELU with the conversion -
    // v_fma_mixlo_f16, ONE rounding - and does not in the row-layout epilogue; the folded-LayerNorm form, the one the towers use, matches)
    const bool ws_epi = epilogue == HGR_EPI_NONE || epilogue == HGR_EPI_BIAS || epilogue == HGR_EPI_BIAS_RELU;
    if (duo_ok && force == 0 && ws_enabled() && !out_f32 && ws_epi && ldc % 8 == 0 && ldc < (1 << 23) && hgr_aligned(C, 16) && (!bias || hgr_aligned(bias, 16)) &&
        ws_covers(M, N, K, WS_PLAIN)) {
        GemmArgs a;
        a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias;
        a.res = nullptr; a.ldr = 0; a.M = M; a.N = N; a.K = K;
        a.m_fastest = ((int64_t)N * K > (int64_t)M * K) ? 1 : 0;
        a.vec_ok = 1; a.dbg = dbg; a.kc = 0; a.csplit = 0; a.group = duo_group();
        launch_ws(a, dtype, WS_PLAIN, epilogue == HGR_EPI_BIAS_RELU ? 2 : 0, epilogue != HGR_EPI_NONE, s);
    }
    else if (duo_ok && (force == 2 || (force == 0 && tduo >= 256 && duo_fits))) launch_d();
    else if (force == 128 || K < 128) launch(0, M, false);
    else if (epi_has_idn16(epilogue) && force != 256) launch(0, M, false);   // only the 128 kernel loads the identity / stores by full lines
    else if (force == 256) launch(0, M, true);
    else if (cost_split < cost_big && cost_split < cost_small) { launch(0, m1, true); launch(m1, M - m1, false); }
    else launch(0, M, cost_big <= cost_small && t256 >= 128);
    HGR_CHECK_LAUNCH("hgr_gemm_nt");
    return HGR_OK;
}

static int conv3x3_launch(const void *x, const void *w, const float *bias, void *out,
                          int B, int H, int W, int C, int Cout, int stride, int Kp, int dtype, bool relu, void *stream) {
    HGR_REQUIRE(x && w && out && (bias || !relu), "hgr_conv3x3_nhwc: null operand");
    HGR_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && (stride == 1 || stride == 2), "hgr_conv3x3_nhwc: bad geometry B=%d H=%d W=%d Cout=%d stride=%d", B, H, W, Cout, stride);
    HGR_REQUIRE(C >= 8 && C % 8 == 0 && C <= 16384, "hgr_conv3x3_nhwc: C=%d must be a multiple of 8 in [8, 16384]", C);
    HGR_REQUIRE(Kp >= 9 * C && Kp % BK == 0, "hgr_conv3x3_nhwc: Kp=%d must be >= 9*C and a multiple of %d", Kp, BK);
    HGR_REQUIRE(hgr_aligned(x, 16) && hgr_aligned(w, 16) && hgr_aligned(out, 8) && (!bias || hgr_aligned(bias, 16)) && Cout % 4 == 0, "hgr_conv3x3_nhwc: misaligned operand / Cout %% 4 != 0");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_conv3x3_nhwc: bad dtype %d", dtype);
    // 32 input channels (the stem at 112 x 112): the direct kernel of hgr_conv_direct.hip; HGR_CONV_DIRECT=0 keeps the implicit GEMM
    static int direct_env = -1;
    if (direct_env < 0) { const char *e = getenv("HGR_CONV_DIRECT"); direct_env = e ? atoi(e) : 1; }
    if (direct_env && C == 32 && stride == 1 && (Cout == 32 || Cout == 64) && Kp >= 288 && hgr_aligned(out, 16))
        return hgr_conv3x3_c32_launch(x, w, bias, out, B, H, W, Cout, Kp, dtype, relu ? 1 : 0, stream, 0, 32);
    // 64 -> 64 channels (layer1 of the ModifiedResNets): the implicit GEMM pulls every input byte nine times through LDS-DMA at
    // ~28 GB/s per CU (258 us at 56 x 56, batch 512); the halo-tile kernel reads it 1.27 times
    if (direct_env && C == 64 && Cout == 64 && stride == 1 && Kp >= 576 && hgr_aligned(out, 16) && (int64_t)B * H * W >= 4096)
        return hgr_conv3x3_c32_launch(x, w, bias, out, B, H, W, Cout, Kp, dtype, relu ? 1 : 0, stream, 0, 64);
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const int64_t M64 = (int64_t)B * Ho * Wo;
    HGR_REQUIRE(M64 < (1ll << 31), "hgr_conv3x3_nhwc: too many output pixels");
    GemmArgs a;
    a.A = (const char *)x; a.lda = 0; a.W = (const char *)w; a.ldw = Kp;
    a.C = out; a.ldc = Cout; a.bias = bias; a.res = nullptr; a.ldr = 0;
    a.M = (int)M64; a.N = Cout; a.K = Kp;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (Cout + BN - 1) / BN;
    a.m_fastest = 0; a.vec_ok = 1; a.dbg = 0; a.kc = 0; a.csplit = 0;
    a.cH = H; a.cW = W; a.cC = C; a.cStride = stride; a.cHo = Ho; a.cWo = Wo;
    a.cMagic = (unsigned)(((1ull << 32) + (unsigned)C - 1) / (unsigned)C);
    a.cUni = (C % 64 == 0) ? 1 : 0;
    // big tiles when the output is at least 256 wide-ish and the launch has >= 160 of them (measured at batch 512: 14x14x256
    // 160 -> 142 us with 392 tiles, 7x7x512 160 -> 135 us with 196 tiles; K is 2304 / 4608 there, so one round is long)
    const int64_t t256 = (int64_t)((a.M + 255) / 256) * ((Cout + 255) / 256);
    // ... and 256-wide tiles do not waste more columns than 128-wide ones would (Cout = 128: half of every 256^2 tile would be
    // padding - measured 1066 vs 646 us at 56x56x128 and 285 vs 160 us at 28x28x128, batch 512)
    const int waste256 = (Cout + 255) / 256 * 256 - Cout, waste128 = (Cout + 127) / 128 * 128 - Cout;
    const bool big = relu && hgr_gemm_force_tile() != 128 && Kp >= 128 && Cout >= 128 && (t256 >= 160 || hgr_gemm_force_tile() == 256) && waste256 <= waste128;
    // 256 x 128 tiles, two workgroups per CU (gemm_nt_duo with the implicit-im2col loader): C % 64 == 0, stride 1, outputs that fill
    // 128-column tiles, operands addressable with 32-bit offsets; HGR_CONV_DUO=0 keeps the kernels above
    static int conv_duo_env = -1;
    if (conv_duo_env < 0) { const char *e = getenv("HGR_CONV_DUO"); conv_duo_env = e ? atoi(e) : 1; }
    const int64_t tduo = (int64_t)((a.M + 255) / 256) * ((Cout + 127) / 128);
    if (conv_duo_env && relu && a.cUni && stride == 1 && Cout % 128 == 0 && Kp == 9 * C && tduo >= 256 && hgr_aligned(out, 16) &&
        (int64_t)B * H * W * C * 2 < (1ll << 31) && (int64_t)Cout * Kp * 2 < (1ll << 32) && hgr_gemm_force_tile() == 0) {
        a.tiles_m = (a.M + 255) / 256; a.tiles_n = Cout / 128; a.group = duo_group();
        a.ln_stats = nullptr; a.ln_flag = nullptr;
        dim3 grid;
        duo_apply_plan(a, true, grid);
        launch_duo(a, dtype, HGR_EPI_BIAS_RELU, false, 5, grid, (hipStream_t)stream);
        HGR_CHECK_LAUNCH("hgr_conv3x3_nhwc");
        return HGR_OK;
    }
    if (big) {
        a.tiles_m = (a.M + 255) / 256; a.tiles_n = (Cout + 255) / 256;
        launch_256(a, dtype, HGR_EPI_BIAS_RELU, false, true, dim3((unsigned)(a.tiles_m * a.tiles_n)), (hipStream_t)stream);
    } else {
        if (Cout <= 64 && a.M >= 1024) {              // tall 256 x 64 tiles: no MFMAs spent on columns that do not exist
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = 1;
            launch_128(a, dtype, relu ? HGR_EPI_BIAS_RELU : HGR_EPI_NONE, false, V128_CONV_TALL, dim3((unsigned)a.tiles_m), (hipStream_t)stream);
            HGR_CHECK_LAUNCH("hgr_conv3x3_nhwc");
            return HGR_OK;
        }
        launch_128(a, dtype, relu ? HGR_EPI_BIAS_RELU : HGR_EPI_NONE, false, V128_CONV, dim3((unsigned)(a.tiles_m * a.tiles_n)), (hipStream_t)stream);
    }
    HGR_CHECK_LAUNCH("hgr_conv3x3_nhwc");
    return HGR_OK;
}

extern "C" int hgr_conv3x3_nhwc(const void *x, const void *w, const float *bias, void *out,
                                int B, int H, int W, int C, int Cout, int stride, int Kp, int dtype, void *stream) {
    return conv3x3_launch(x, w, bias, out, B, H, W, C, Cout, stride, Kp, dtype, true, stream);
}

extern "C" int hgr_conv3x3_nhwc_plain(const void *x, const void *w, void *out, int B, int H, int W, int C, int Cout, int Kp,
                                      int dtype, void *stream) {
    return conv3x3_launch(x, w, nullptr, out, B, H, W, C, Cout, 1, Kp, dtype, false, stream);
}

extern "C" int hgr_gemm_nt_splitk(const void *A, int64_t lda, const void *W, int64_t ldw, float *partial, int64_t ldc,
                                  int M, int N, int K, int kc, int dtype, void *stream) {
    HGR_REQUIRE(A && W && partial, "hgr_gemm_nt_splitk: null operand");
    HGR_REQUIRE(M >= 1 && N >= 1 && K >= BK && K % BK == 0 && kc >= BK && kc % BK == 0, "hgr_gemm_nt_splitk: bad shape M=%d N=%d K=%d kc=%d", M, N, K, kc);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && hgr_aligned(A, 16) && hgr_aligned(W, 16), "hgr_gemm_nt_splitk: operands must be 16-byte aligned with leading dimensions %% 8 == 0");
    HGR_REQUIRE(ldc >= N && ldc % 4 == 0 && hgr_aligned(partial, 16), "hgr_gemm_nt_splitk: partial must be 16-byte aligned, ldc %% 4 == 0");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_gemm_nt_splitk: bad dtype %d", dtype);
    const int S = (K + kc - 1) / kc;
    HGR_REQUIRE(S <= 65535, "hgr_gemm_nt_splitk: %d splits exceed the grid limit", S);
    GemmArgs a;
    a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw;
    a.C = partial; a.ldc = ldc; a.bias = nullptr; a.res = nullptr; a.ldr = 0;
    a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    a.m_fastest = 0; a.vec_ok = 1; a.dbg = 0; a.kc = kc; a.csplit = (int64_t)M * ldc;
    a.cH = a.cW = a.cC = a.cStride = a.cHo = a.cWo = 0; a.cMagic = 0; a.cUni = 0;
    // every slice at least 2 K-tiles deep and an output of at least one 256^2 tile: the deep-pipelined kernel
    const bool big = hgr_gemm_force_tile() != 128 && M >= 256 && N >= 256 && kc >= 128 && (K - (S - 1) * kc) >= 128;
    if (big) {
        a.tiles_m = (M + 255) / 256; a.tiles_n = (N + 255) / 256;
        launch_256(a, dtype, HGR_EPI_NONE, true, false, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)S), (hipStream_t)stream);
        HGR_CHECK_LAUNCH("hgr_gemm_nt_splitk");
        return HGR_OK;
    }
    launch_128(a, dtype, HGR_EPI_NONE, true, V128_PLAIN, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)S), (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_splitk");
    return HGR_OK;
}

// ---- LayerNorm folded into the GEMMs around it --------------------------------------------------------------------------
namespace {
int ln_common_checks(const char *who, const void *A, int64_t lda, const void *W, int64_t ldw, int M, int N, int K, int dtype) {
    HGR_REQUIRE(A && W, "%s: null operand", who);
    HGR_REQUIRE(M >= 1 && N >= 128 && N % 128 == 0 && K >= 128 && K % 64 == 0, "%s: bad shape M=%d N=%d K=%d (N %% 128 == 0, K %% 64 == 0, K >= 128)", who, M, N, K);
    HGR_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && hgr_aligned(A, 16) && hgr_aligned(W, 16), "%s: operands must be 16-byte aligned with leading dimensions %% 8 == 0", who);
    HGR_REQUIRE((int64_t)M * lda * 2 < (1ll << 32) && (int64_t)N * ldw * 2 < (1ll << 32) && N < (1 << 20), "%s: operands beyond 4 GB / N >= 2^20", who);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "%s: bad dtype %d", who, dtype);
    return HGR_OK;
}
void ln_args(GemmArgs &a, const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc, int M, int N, int K) {
    a.A = (const char *)A; a.lda = lda; a.W = (const char *)W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = nullptr; a.res = nullptr; a.ldr = 0;
    a.M = M; a.N = N; a.K = K; a.tiles_m = (M + 255) / 256; a.tiles_n = N / 128;
    a.m_fastest = ((int64_t)N * K > (int64_t)M * K) ? 1 : 0; a.vec_ok = 1; a.dbg = duo_dbg(); a.kc = 0; a.csplit = 0; a.group = duo_group();
    a.cH = a.cW = a.cC = a.cStride = a.cHo = a.cWo = 0; a.cMagic = 0; a.cUni = 0;
    a.ln_stats = nullptr; a.ln_slots = 0; a.ln_eps = 0.f; a.ln_xh = a.ln_xl = nullptr; a.ln_ldx = 0; a.ln_s = a.ln_c = nullptr; a.ln_flag = nullptr; a.ln_guard = 0.f;
}
}  // namespace

extern "C" int hgr_gemm_nt_res_stats(const void *A, int64_t lda, const void *W, int64_t ldw, void *xh, void *xl, int64_t ldx,
                                     const float *bias, float *stats, int M, int N, int K, int dtype, void *stream) {
    return hgr_gemm_nt_res_stats_guard(A, lda, W, ldw, xh, xl, ldx, bias, stats, 0.f, nullptr, M, N, K, dtype, stream);
}

extern "C" int hgr_gemm_nt_res_stats_guard(const void *A, int64_t lda, const void *W, int64_t ldw, void *xh, void *xl, int64_t ldx,
                                           const float *bias, float *stats, float guard_sumsq, uint32_t *flag,
                                           int M, int N, int K, int dtype, void *stream) {
    if (int rc = ln_common_checks("hgr_gemm_nt_res_stats", A, lda, W, ldw, M, N, K, dtype)) return rc;
    HGR_REQUIRE(!flag || (guard_sumsq > 0.f && hgr_aligned(flag, 4)), "hgr_gemm_nt_res_stats_guard: flag needs guard_sumsq > 0 and 4-byte alignment");
    HGR_REQUIRE(xh && xl && bias && stats, "hgr_gemm_nt_res_stats: null xh / xl / bias / stats");
    // the interior epilogue reads and writes the pair with 16-byte accesses (8 columns per lane) at 32-bit offsets of up to 127 rows
    HGR_REQUIRE(ldx >= N && ldx % 8 == 0 && ldx < (1 << 23) && hgr_aligned(xh, 16) && hgr_aligned(xl, 16), "hgr_gemm_nt_res_stats: xh / xl must be 16-byte aligned, ldx %% 8 == 0, ldx < 2^23");
    HGR_REQUIRE(hgr_aligned(bias, 16) && hgr_aligned(stats, 8), "hgr_gemm_nt_res_stats: bias must be 16-byte, stats 8-byte aligned");
    GemmArgs a;
    ln_args(a, A, lda, W, ldw, nullptr, 0, M, N, K);
    a.bias = bias;
    a.ln_stats = stats; a.ln_slots = N / 64; a.ln_xh = xh; a.ln_xl = xl; a.ln_ldx = ldx;
    a.ln_flag = flag; a.ln_guard = guard_sumsq;
    if (ws_enabled() && hgr_gemm_force_tile() == 0 && ws_covers(M, N, K, WS_LNP)) {
        launch_ws(a, dtype, WS_LNP, 0, true, (hipStream_t)stream);
        HGR_CHECK_LAUNCH("hgr_gemm_nt_res_stats");
        return HGR_OK;
    }
    dim3 grid;
    duo_apply_plan(a, true, grid);
    launch_duo(a, dtype, HGR_EPI_BIAS_RESIDUAL, true, 1, grid, (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_res_stats");
    return HGR_OK;
}

extern "C" int hgr_gemm_nt_ln(const void *X16, int64_t ldx, const void *Wfold, int64_t ldw, void *C, int64_t ldc,
                              const float *ln_s, const float *ln_c, const float *stats, float eps,
                              int M, int N, int K, int dtype, int act, void *stream) {
    if (int rc = ln_common_checks("hgr_gemm_nt_ln", X16, ldx, Wfold, ldw, M, N, K, dtype)) return rc;
    HGR_REQUIRE(C && ln_s && ln_c && stats, "hgr_gemm_nt_ln: null C / ln_s / ln_c / stats");
    HGR_REQUIRE(K % 128 == 0, "hgr_gemm_nt_ln: the row width K=%d must be a multiple of 128 (two 64-column statistic slots per 16-byte load)", K);
    // 16-bit C rows are addressed as a 64-bit tile base + a 32-bit per-lane byte offset of up to 127 rows: 127 * ldc * 2 < 2^32
    HGR_REQUIRE(ldc >= N && ldc % 8 == 0 && ldc < (1 << 23) && hgr_aligned(C, 16), "hgr_gemm_nt_ln: C must be 16-byte aligned with ldc %% 8 == 0, ldc < 2^23");
    HGR_REQUIRE(hgr_aligned(ln_s, 16) && hgr_aligned(ln_c, 16) && hgr_aligned(stats, 16), "hgr_gemm_nt_ln: ln_s / ln_c / stats must be 16-byte aligned");
    HGR_REQUIRE(act == 0 || act == 1, "hgr_gemm_nt_ln: act must be 0 (none) or 1 (QuickGELU), got %d", act);
    GemmArgs a;
    ln_args(a, X16, ldx, Wfold, ldw, C, ldc, M, N, K);
    a.ln_stats = const_cast<float *>(stats); a.ln_slots = K / 64; a.ln_eps = eps; a.ln_s = ln_s; a.ln_c = ln_c;
    if (hgr_gemm_force_tile() == 0 && p8_wanted(M, N, K)) {
        launch_p8(a, dtype, act, (hipStream_t)stream);
        HGR_CHECK_LAUNCH("hgr_gemm_nt_ln");
        return HGR_OK;
    }
    if (ws_enabled() && hgr_gemm_force_tile() == 0 && ws_covers(M, N, K, WS_LNC)) {
        launch_ws(a, dtype, WS_LNC, act, false, (hipStream_t)stream);
        HGR_CHECK_LAUNCH("hgr_gemm_nt_ln");
        return HGR_OK;
    }
    dim3 grid;
    duo_apply_plan(a, true, grid);
    launch_duo(a, dtype, act ? HGR_EPI_BIAS_QUICKGELU : HGR_EPI_BIAS, false, 2, grid, (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_ln");
    return HGR_OK;
}

extern "C" int hgr_gemm_nt_bias_gelu_dual(const void *A, int64_t lda, const void *W, int64_t ldw, void *pre, int64_t ldpre, void *post, int64_t ldpost,
                                          const float *bias, int M, int N, int K, int dtype, void *stream) {
    if (int rc = ln_common_checks("hgr_gemm_nt_bias_gelu_dual", A, lda, W, ldw, M, N, K, dtype)) return rc;
    HGR_REQUIRE(pre && post && bias, "hgr_gemm_nt_bias_gelu_dual: null pre / post / bias");
    HGR_REQUIRE(ldpre >= N && ldpost >= N && ldpre % 8 == 0 && ldpost % 8 == 0 && ldpre < (1 << 23) && ldpost < (1 << 23) && hgr_aligned(pre, 16) && hgr_aligned(post, 16) && hgr_aligned(bias, 16),
                "hgr_gemm_nt_bias_gelu_dual: pre / post / bias must be 16-byte aligned, leading dimensions >= N, %% 8 == 0, < 2^23");
    GemmArgs a;
    ln_args(a, A, lda, W, ldw, pre, ldpre, M, N, K);
    a.bias = bias; a.ln_xh = post; a.ln_ldx = ldpost;
    dim3 grid;
    duo_apply_plan(a, true, grid);
    launch_duo(a, dtype, HGR_EPI_BIAS, false, 4, grid, (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_bias_gelu_dual");
    return HGR_OK;
}

extern "C" int hgr_gemm_nt_qgelu_grad_colsum(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc, const void *pre, int64_t ldpre,
                                             float *colsum_part, int M, int N, int K, int dtype, void *stream) {
    if (int rc = ln_common_checks("hgr_gemm_nt_qgelu_grad_colsum", A, lda, W, ldw, M, N, K, dtype)) return rc;
    HGR_REQUIRE(C && pre && colsum_part, "hgr_gemm_nt_qgelu_grad_colsum: null C / pre / colsum_part");
    HGR_REQUIRE(ldc >= N && ldpre >= N && ldc % 8 == 0 && ldpre % 8 == 0 && ldc < (1 << 20) && ldpre < (1 << 20) && hgr_aligned(C, 16) && hgr_aligned(pre, 16) && hgr_aligned(colsum_part, 16),
                "hgr_gemm_nt_qgelu_grad_colsum: C / pre / colsum_part must be 16-byte aligned, leading dimensions >= N, %% 8 == 0, < 2^20");
    GemmArgs a;
    ln_args(a, A, lda, W, ldw, C, ldc, M, N, K);
    a.res = (const float *)pre; a.ldr = ldpre;
    a.colsum = colsum_part; a.colsum_units = (M + 63) / 64;
    dim3 grid;
    duo_apply_plan(a, true, grid);
    launch_duo(a, dtype, HGR_EPI_QGELU_GRAD16, false, 0, grid, (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_gemm_nt_qgelu_grad_colsum");
    return HGR_OK;
}

// ---- logits GEMM with the evaluation consumers in its epilogue ------------------------------------------------------------
#define HGR_LE_PARAMS const void *feat16, const void *zsl_perm16, int rows, int D, int n_perm, const int32_t *tpos_perm, const int32_t *epos_perm, \
    const int32_t *level_first, int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train, const int32_t *test_cols, int n_test, int k, \
    int32_t *out_level, int32_t *out_top1, int32_t *out_topk, void *workspace, int dtype, void *stream
#define HGR_LE_ARGS feat16, zsl_perm16, rows, D, n_perm, tpos_perm, epos_perm, level_first, n_levels, filler_pos, train_cols, n_train, test_cols, n_test, k, \
    out_level, out_top1, out_topk, workspace, dtype, stream
static int logits_eval_stages(int stages, HGR_LE_PARAMS);
extern "C" int hgr_logits_eval(HGR_LE_PARAMS) { return logits_eval_stages(3, HGR_LE_ARGS); }
// the two stages on their own (same arguments; the row stage reads the workspace the tile stage of the same call sequence wrote):
// bench.py times them separately, hgr_logits_eval is the pair
extern "C" int hgr_logits_eval_tile_stage(HGR_LE_PARAMS) { return logits_eval_stages(1, HGR_LE_ARGS); }
extern "C" int hgr_logits_eval_row_stage(HGR_LE_PARAMS) { return logits_eval_stages(2, HGR_LE_ARGS); }

int hgr_logits_eval_rows_launch(const void *feat, const void *zslp, int D, int S, const unsigned long long *keys, const float *tmax,
                                const int *gp1, const float *gm2, const int32_t *level_first, int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                                const int32_t *epos, const int32_t *test_cols, int n_test, int k, int32_t *out_level, int32_t *out_top1,
                                int32_t *out_topk, int rows, int dtype, void *stream);      // hgr_select.hip

extern "C" int64_t hgr_logits_eval_workspace_bytes(int rows, int n_perm) {
    if (rows < 1 || n_perm < 96 || n_perm % 96) return -1;
    return (int64_t)rows * (n_perm / 32) * 32;           // per (row, 32-column slice): 8-byte train key + per 16-column group (max, position, second) = 2 x 12 bytes
}

static int logits_eval_stages(int stages, const void *feat16, const void *zsl_perm16, int rows, int D, int n_perm,
                               const int32_t *tpos_perm, const int32_t *epos_perm, const int32_t *level_first,
                               int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                               const int32_t *test_cols, int n_test, int k,
                               int32_t *out_level, int32_t *out_top1, int32_t *out_topk, void *workspace, int dtype, void *stream) {
    HGR_REQUIRE(feat16 && zsl_perm16 && tpos_perm && epos_perm && level_first && filler_pos && train_cols && out_level && workspace, "hgr_logits_eval: null operand");
    HGR_REQUIRE(rows >= 1 && D >= 128 && D % 128 == 0 && D <= 1024, "hgr_logits_eval: rows=%d D=%d unsupported (D %% 128 == 0, D <= 1024)", rows, D);
    HGR_REQUIRE(n_perm >= 96 && n_perm % 96 == 0 && n_perm / 32 <= 1024, "hgr_logits_eval: n_perm=%d must be a multiple of 96 and <= 32768 (32-column level-aligned slices, 96-column slabs)", n_perm);
    HGR_REQUIRE(n_levels >= 1 && n_levels <= 32 && n_train >= 1, "hgr_logits_eval: bad sizes (n_levels <= 32)");
    HGR_REQUIRE(k == 0 || (out_topk && test_cols && k >= 1 && k <= 32 && n_test >= k), "hgr_logits_eval: bad top-k arguments");
    HGR_REQUIRE(hgr_aligned(feat16, 16) && hgr_aligned(zsl_perm16, 16) && hgr_aligned(tpos_perm, 16) && hgr_aligned(epos_perm, 16) && hgr_aligned(workspace, 16), "hgr_logits_eval: operands must be 16-byte aligned");
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "hgr_logits_eval: bad dtype %d", dtype);
    HGR_REQUIRE((int64_t)rows * D * 2 < (1ll << 32) && (int64_t)n_perm * D * 2 < (1ll << 32), "hgr_logits_eval: operands beyond 4 GB");
    const int S = n_perm / 32;
    SlabArgs a;
    a.A = (const char *)feat16; a.lda = D; a.W = (const char *)zsl_perm16; a.ldw = D; a.M = rows; a.K = D; a.Np = n_perm;
    a.tpos = tpos_perm; a.epos = epos_perm; a.S = S;
    { static int d = -1; if (d < 0) d = hgr_lab_env("HGR_LS_DBG"); a.dbg = d; }
    a.ev_key = (unsigned long long *)workspace;
    a.ev_tmax = (float *)((char *)workspace + (size_t)rows * S * 8);
    a.ev_p1 = (int *)((char *)workspace + (size_t)rows * S * 16);
    a.ev_m2 = (float *)((char *)workspace + (size_t)rows * S * 24);
    if (stages & 1) launch_logits_slab(a, dtype, (hipStream_t)stream);
    HGR_CHECK_LAUNCH("hgr_logits_eval (tile stage)");
    if (!(stages & 2)) return HGR_OK;
    return hgr_logits_eval_rows_launch(feat16, zsl_perm16, D, S, a.ev_key, a.ev_tmax, a.ev_p1, a.ev_m2, level_first, n_levels, filler_pos, train_cols, n_train,
                                       epos_perm, test_cols, n_test, k, out_level, out_top1, out_topk, rows, dtype, stream);
}

