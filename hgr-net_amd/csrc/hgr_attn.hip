// hgr_mha: softmax(q k^T / sqrt(d) [+ causal mask]) v for short sequences, d_head = 64.
//
// Sequences on this path are 50 (ViT-B/32, RN50 attnpool), <= 77 (text, usually trimmed to <= 16)
// or 257 (ViT-L/14) tokens, so one workgroup holds all keys/values of one (batch, head) in LDS and
// the whole score row of a query lives in registers: no online softmax, no rescaling.
//
// One 256-thread workgroup per (batch, head); wave w takes query tiles w, w+4, ... of 16 queries.
//   S^T = K . Q^T   (v_mfma 16x16x32, K rows as the A operand from LDS, Q rows as the B operand
//                    straight from global memory): lane (q = lane&15, g = lane>>4) then holds
//                    S[q][key = 16t + 4g + e] in acc[t][e] - the query is on the lane, so the
//                    softmax reduction is in-register plus two shuffles (xor 16, 32);
//   O^T = V^T . P^T : the exponentiated accumulators ARE the B operand of the second product after
//                    a 16-bit convert (accumulator-as-operand, cdna_hip_programming.md section 3):
//                    k-slot j of lane-group g in 32-key step s is key 32s + 16(j>>2) + 4g + (j&3),
//                    so the A operand reads V^T with the same key permutation: two transposing LDS reads
//                    (ds_read_b64_tr_b16) of the row-major V tile - V is staged as it lies in memory, no
//                    transposed image is written.  No LDS round trip for P.
// Output lane layout: 4 consecutive head dims of one query -> 8-byte stores.
// HBM-bound in practice (reads 3W, writes W 16-bit values per token), the MFMA work is ~1 % of a layer.
#include "hgr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short mha_s16x4;
typedef __attribute__((ext_vector_type(8))) short mha_s16x8;

template <int DT, int KT, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void mha_fwd(const typename T16<DT>::elem *__restrict__ qkv,
                                               typename T16<DT>::elem *__restrict__ out, int L, int H, float2 *__restrict__ stats, int QL) {
    typedef typename T16<DT>::elem E;
    typedef typename T16<DT>::vec8 vec8;
    typedef typename T16<DT>::vec4 vec4;
    constexpr int LP = KT * 32;        // keys padded to the MFMA k-step
    // V^T fragments: for short sequences (<= 64 keys: ViT-B/32, the attention pool, prompts) V is staged as it lies in memory and the
    // fragments come out of ds_read_b64_tr_b16 (29.5 vs 35 us at ViT-B/32 batch 512: no scalar transposing LDS stores on the
    // kernel's one critical path); for long sequences the transposed image wins (ViT-L/14, 257 keys: 646 vs 710 us at batch 512 -
    // the staging is amortised over 17 query tiles, the plain 8-byte reads are cheaper than the transposing ones)
    constexpr bool VTR = KT <= 2;
    constexpr int VR = 72;             // VTR: V row stride in elements (144 B: the 4-row x 32-byte blocks of a transposing read spread over the banks)
    constexpr int VS = LP + 4;         // else: V^T row stride in elements (8-byte aligned rows, spreads banks)
    __shared__ __attribute__((aligned(16))) char sK[LP * 128];
    __shared__ __attribute__((aligned(16))) E sV[VTR ? LP * VR : 64 * VS];
    E *const sVt = sV;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int W = H * 64;
    const int64_t ld = 3 * (int64_t)W;
    const E *base = qkv + (int64_t)b * L * ld + h * 64;

    // this wave's first query tile is requested together with K / V: one memory round trip per workgroup, not two (with
    // L <= 64 - ViT-B/32, the attention pool, trimmed prompts - a wave has exactly one tile and the kernel is one round trip)
    // QL = number of leading query rows whose output is wanted (L: all; 1: the class token of a ViT's last block, the only row its
    // head reads): query tiles beyond QL are neither loaded nor computed, rows beyond it are not stored; the rows that are stored
    // go through exactly the same instructions on the same data
    vec8 q0n, q1n;
    {
        const int qrow = min(wave * 16 + r, QL - 1);       // rows beyond QL are never stored: their lanes re-read row QL - 1 (a cache hit)
        q0n = *(const vec8 *)(base + qrow * ld + g * 8);
        q1n = *(const vec8 *)(base + qrow * ld + 32 + g * 8);
    }
    // stage K (swizzled rows) and V^T (zero-filled past L: 0 * garbage must stay 0).  LP * 8 = KT * 256 chunks of 16 bytes: exactly KT per
    // thread.  ALL 2 KT loads are issued before the first LDS store: written as one loop over the chunks, hipcc kept it a loop (the trip
    // count depends on tid as far as it can tell) with the two loads of an iteration waited for before its stores - KT serial round
    // trips of global memory per workgroup, nine at 257 keys, which was most of this kernel's time there.
    {
        vec8 kv[KT], vv[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const int idx = tid + 256 * it;
            const int rc = min(idx >> 3, L - 1);           // unconditional loads of a clamped row (a load under a branch is waited for at once)
            kv[it] = *(const vec8 *)(base + rc * ld + W + (idx & 7) * 8);
            vv[it] = *(const vec8 *)(base + rc * ld + 2 * W + (idx & 7) * 8);
        }
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const int idx = tid + 256 * it;
            const int row = idx >> 3, c = idx & 7;
            if (row >= L)
#pragma unroll
                for (int e = 0; e < 8; ++e) { kv[it][e] = (E)0.f; vv[it][e] = (E)0.f; }
            *(vec8 *)(sK + row * 128 + ((c ^ (row & 7)) * 16)) = kv[it];
            if (VTR) *(vec8 *)(sV + row * VR + c * 8) = vv[it];
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) sVt[(c * 8 + e) * VS + row] = vv[it][e];
        }
    }
    __syncthreads();

    const int sw0 = ((0 + g) ^ (r & 7)) * 16, sw1 = ((4 + g) ^ (r & 7)) * 16;
    for (int qt = wave; qt * 16 < QL; qt += 4) {
        const int q = qt * 16 + r;
        const vec8 q0 = q0n, q1 = q1n;
        if ((qt + 4) * 16 < QL) {                  // the next tile's queries travel while this tile is computed
            const int qrow = min(q + 64, QL - 1);
            q0n = *(const vec8 *)(base + qrow * ld + g * 8);
            q1n = *(const vec8 *)(base + qrow * ld + 32 + g * 8);
        }

        f32x4 acc[2 * KT];
#pragma unroll
        for (int t = 0; t < 2 * KT; ++t) {
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const char *kr = sK + (t * 16 + r) * 128;
            acc[t] = T16<DT>::mfma16(*(const vec8 *)(kr + sw0), q0, acc[t]);
            acc[t] = T16<DT>::mfma16(*(const vec8 *)(kr + sw1), q1, acc[t]);
            // long sequences: without a fence every four key tiles the scheduler hoists ALL K fragments above the first MFMA (KT = 9:
            // 438 registers -> one wave per SIMD, one workgroup per CU); with it 135 - 200 registers, two workgroups per CU
            if (KT >= 5 && (t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * KT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = t * 16 + g * 4 + e;
                float s = acc[t][e] * 0.125f;     // 64^-0.5, exact
                if (key >= L || (CAUSAL && key > q)) s = -INFINITY;
                acc[t][e] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        // e^(s - mx) as 2^(s log2e - mx log2e): one FMA + v_exp_f32 per element where subtract, multiply, v_exp_f32 were three (round 6;
        // qkv_attn evaluates the same expression)
        const float nmx = -mx * 1.4426950408889634f;
#pragma unroll
        for (int t = 0; t < 2 * KT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pexp = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[t][e], 1.4426950408889634f, nmx));
                acc[t][e] = pexp;
                sum += pexp;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // training: the row's softmax statistics (max of the scaled scores, 1 / sum of exponentials) for hgr_mha_bwd_stats
        if (stats && g == 0 && q < QL) stats[((int64_t)b * H + h) * L + q] = make_float2(mx, inv);

        f32x4 o[4];
#pragma unroll
        for (int td = 0; td < 4; ++td) o[td] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            vec8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pf[e] = (E)acc[2 * s][e]; pf[4 + e] = (E)acc[2 * s + 1][e]; }
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                // A operand = V^T[d = 16 td + r][keys 32 s + 4 g .. + 3, 32 s + 16 + 4 g .. + 3]: lane 4 q + pp of the 16-lane group g
                // supplies (key row 32 s + 4 g + q, columns 16 td + 4 pp ..) and receives the 4 keys of column 16 td + (lane & 15)
                vec8 vf;
                if (VTR) {
                    const E *vb = sV + (s * 32 + g * 4 + (r >> 2)) * VR + td * 16 + (r & 3) * 4;
                    const mha_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 mha_s16x4 *)vb);
                    const mha_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 mha_s16x4 *)(vb + 16 * VR));
                    vf = __builtin_bit_cast(vec8, (mha_s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                } else {
                    const E *vr = sVt + (td * 16 + r) * VS + s * 32 + g * 4;
                    const vec4 lo = *(const vec4 *)vr;
                    const vec4 hi = *(const vec4 *)(vr + 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                }
                o[td] = T16<DT>::mfma16(vf, pf, o[td]);
            }
            if (KT >= 5) __builtin_amdgcn_sched_barrier(0);       // likewise for the V^T fragments
        }
        if (q < QL) {
            E *orow = out + ((int64_t)b * L + q) * W + h * 64 + g * 4;
#pragma unroll
            for (int td = 0; td < 4; ++td)
                *(u32x2 *)(orow + td * 16) = (u32x2){mul_pack16<DT>(o[td][0], inv, o[td][1], inv), mul_pack16<DT>(o[td][2], inv, o[td][3], inv)};   // product rounded ONCE to f16 (hgr_gemm_common.h): qkv_attn's twin
        }
    }
}

template <int DT, int KT>
void launch_kt(const void *qkv, void *out, int B, int L, int H, bool causal, hipStream_t s, float2 *stats, int ql) {
    typedef typename T16<DT>::elem E;
    dim3 grid(B * H), block(256);
    if (causal) hipLaunchKernelGGL((mha_fwd<DT, KT, true>), grid, block, 0, s, (const E *)qkv, (E *)out, L, H, stats, ql);
    else hipLaunchKernelGGL((mha_fwd<DT, KT, false>), grid, block, 0, s, (const E *)qkv, (E *)out, L, H, stats, ql);
}

template <int DT>
void launch_dt(const void *qkv, void *out, int B, int L, int H, bool causal, hipStream_t s, float2 *stats, int ql) {
    const int kt = (L + 31) / 32;
    if (kt <= 1) launch_kt<DT, 1>(qkv, out, B, L, H, causal, s, stats, ql);
    else if (kt <= 2) launch_kt<DT, 2>(qkv, out, B, L, H, causal, s, stats, ql);
    else if (kt <= 3) launch_kt<DT, 3>(qkv, out, B, L, H, causal, s, stats, ql);
    else if (kt <= 5) launch_kt<DT, 5>(qkv, out, B, L, H, causal, s, stats, ql);
    else launch_kt<DT, 9>(qkv, out, B, L, H, causal, s, stats, ql);
}

}  // namespace

static int mha_entry(const char *name, const void *qkv, void *out, float *stats, int B, int L, int heads, int causal, int dtype, void *stream, int q_rows = 0) {
    HGR_REQUIRE(qkv && out, "%s: null operand", name);
    HGR_REQUIRE(B >= 1 && heads >= 1 && L >= 1 && L <= 288, "%s: B=%d heads=%d L=%d unsupported (1 <= L <= 288)", name, B, heads, L);
    HGR_REQUIRE((int64_t)B * heads < (1ll << 31), "%s: grid too large", name);
    HGR_REQUIRE(hgr_aligned(qkv, 16) && hgr_aligned(out, 16) && hgr_aligned(stats, 8), "%s: operands must be 16-byte aligned (stats: 8)", name);
    HGR_REQUIRE(dtype == HGR_BF16 || dtype == HGR_F16, "%s: bad dtype %d", name, dtype);
    HGR_REQUIRE(q_rows >= 0 && q_rows <= L, "%s: q_rows=%d must lie in [1, L]", name, q_rows);
    const int ql = q_rows ? q_rows : L;
    if (dtype == HGR_BF16) launch_dt<HGR_BF16>(qkv, out, B, L, heads, causal != 0, (hipStream_t)stream, (float2 *)stats, ql);
    else launch_dt<HGR_F16>(qkv, out, B, L, heads, causal != 0, (hipStream_t)stream, (float2 *)stats, ql);
    HGR_CHECK_LAUNCH(name);
    return HGR_OK;
}

extern "C" int hgr_mha(const void *qkv, void *out, int B, int L, int heads, int causal, int dtype, void *stream) {
    return mha_entry("hgr_mha", qkv, out, nullptr, B, L, heads, causal, dtype, stream);
}

extern "C" int hgr_mha_rows(const void *qkv, void *out, int B, int L, int heads, int causal, int q_rows, int dtype, void *stream) {
    HGR_REQUIRE(q_rows >= 1, "hgr_mha_rows: q_rows=%d must be >= 1", q_rows);
    return mha_entry("hgr_mha_rows", qkv, out, nullptr, B, L, heads, causal, dtype, stream, q_rows);
}

extern "C" int hgr_mha_stats(const void *qkv, void *out, float *stats, int B, int L, int heads, int causal, int dtype, void *stream) {
    HGR_REQUIRE(stats, "hgr_mha_stats: null stats");
    return mha_entry("hgr_mha_stats", qkv, out, stats, B, L, heads, causal, dtype, stream);
}
