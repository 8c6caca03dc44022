/*
 * libhgr.so - C ABI of the MI355X (gfx950) kernels under the HGR-Net zero-shot forward path.
 *
 * The reference (WilliamYi96/HGR-Net) is 100 % Python on stock PyTorch and has no FFI of its own;
 * its de-facto boundary is the Python call surface between main.py and the model object
 * (SURVEY.md section 8b).  This header is the boundary *underneath* that surface: every entry point
 * replaces one stock-PyTorch call site of the reference, cited per function as file:line relative to
 * the reference checkout.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every tensor argument is a raw DEVICE pointer, row-major;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is asynchronous
 *     and stream-ordered, never synchronises and allocates no device memory.  The only process-wide mutable state is
 *     (a) the development knobs hgr_gemm_set_tile / hgr_gemm_set_tail / hgr_gemm_set_persist (plan overrides for A/B runs and tests) and
 *     (b) the optional RCCL communicator created / destroyed explicitly by hgr_comm_init / hgr_comm_destroy and
     (c) the 16-slot scratch ring of hgr_sumsq (see there);
 *   - returns 0 on success, a negative HGR_E* code otherwise; hgr_last_error() returns the message of
 *     the last failure on the calling thread.  Arguments are validated on the host before any launch
 *     (shape / alignment assumptions of the kernel), so a bad call fails loudly instead of faulting;
 *   - `dtype` selects the 16-bit MFMA input type of an op: HGR_BF16 or HGR_F16 (same MFMA rate on
 *     gfx950).  Accumulation, LayerNorm, softmax and the residual stream are always fp32.
 */
#ifndef HGR_H
#define HGR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round-2 additions (LayerNorm-folded GEMMs, hgr_logits_eval, RCCL collectives, the training-step fusions); 4: round-4 additions
 * (hgr_gemm_nt_ln_mha); every earlier entry point is unchanged */
#define HGR_ABI_VERSION 4

enum { HGR_OK = 0, HGR_EINVAL = -1, HGR_EUNSUPPORTED = -2, HGR_ELAUNCH = -3 };

typedef enum { HGR_BF16 = 0, HGR_F16 = 1 } hgr_dtype_t;

/* epilogues of hgr_gemm_nt */
typedef enum {
    HGR_EPI_NONE = 0,           /* C = A W^T                                                            */
    HGR_EPI_BIAS = 1,           /* C = A W^T + bias                  (nn.Linear; in_proj of MHA)        */
    HGR_EPI_BIAS_QUICKGELU = 2, /* C = g(A W^T + bias), g(x) = x*sigmoid(1.702x)  (clip/model.py:162-164,177-180) */
    HGR_EPI_BIAS_RESIDUAL = 3,  /* C = residual + A W^T + bias       (x + attn(..), x + mlp(..): clip/model.py:186-187) */
    HGR_EPI_BIAS_RELU = 4,      /* C = relu(A W^T + bias)            (conv + folded BN + ReLU: clip/model.py:43-44,137-138) */
    HGR_EPI_BIAS_ADD16_RELU = 5,/* C = relu(A W^T + bias + identity) (bn3(conv3) ; out += identity ; relu: clip/model.py:46-52);
                                   `residual` points at 16-bit values [M, ldr], 16-bit output only */
    HGR_EPI_ACCUM = 6,          /* C += A W^T, fp32 C                (weight gradients dW += dY^T X of loss.backward(), clip_tree.py:276) */
    HGR_EPI_QGELU_GRAD16 = 7    /* C = (A W^T) * g'(pre), g' = d/dx x*sigmoid(1.702x); pre = 16-bit [M, ldr] in `residual`, 16-bit C, no bias:
                                   the gradient w.r.t. the QuickGELU pre-activation straight out of the c_proj data-gradient GEMM
                                   (autograd of clip/model.py:162-164,177-180 behind clip_tree.py:276,280) */
} hgr_epilogue_t;

int hgr_abi_version(void);
const char *hgr_last_error(void);

/*
 * C[M,N] = epilogue(A[M,K] . W[N,K]^T), fp32 accumulate on MFMA.
 * Replaces F.linear / nn.Linear / `@` at clip/model.py:171,183 (in/out projection inside
 * nn.MultiheadAttention), :177-180 (mlp.c_fc, mlp.c_proj), :207,220-222 (patch conv as GEMM over
 * disjoint patches), :234 (x @ proj), :350 (@ text_projection) and the logits product
 * `feats @ zsl_weights.T` at model/clip_tree.py:331.
 *   A, W      16-bit (`dtype`), leading dimensions lda/ldw in elements, both K-contiguous ("NT")
 *   C         fp32 if out_f32 else 16-bit (`dtype`), leading dimension ldc
 *   bias      fp32 [N] or NULL;  residual fp32 [M, ldr] (16-bit for HGR_EPI_BIAS_ADD16_RELU / HGR_EPI_QGELU_GRAD16) or NULL;
 *             may alias C when element sizes match
 * Requirements: K % 64 == 0; lda, ldw % 8 == 0; A, W 16-byte aligned; M, N >= 1.
 */
int hgr_gemm_nt(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                const float *bias, const void *residual, int64_t ldr,
                int M, int N, int K, int dtype, int epilogue, int out_f32, void *stream);

/*
 * Tile plan override for hgr_gemm_nt: 0 = cost model (default), 128 = the 128x128 kernel only, 256 = the 256x256
 * kernel wherever it is legal (K >= 128).  Process-wide; the HGR_GEMM_TILE environment variable sets the
 * initial value.  Results do not depend on the plan (every tile sums K in the same order); parity tests run every
 * shape under each plan.  Returns the previous value, or a negative HGR_E* code for any other argument.
 */
int hgr_gemm_set_tile(int tile);

/*
 * Tail plan of the 256 x 128 tile kernel (hgr_gemm_nt and the LayerNorm-folded forms below): a launch whose tile count is not a
 * whole number of rounds of the chip's 512 workgroup slots gives its LAST row panels to 128 x 128 half tiles, dispatched after
 * the full tiles, so that the last round is not left to a few workgroups ("tail plan").  enabled = 1 (default; HGR_DUO_TAIL)
 * / 0; full_panels = -1 lets the host's list-scheduling model choose how many 256-row panels stay on full tiles, n >= 0 forces
 * it (measurement sweeps; HGR_DUO_PB).  Every output element sums K in the same order on either tile: results are bit-identical.
 * Process-wide development knob like hgr_gemm_set_tile.  Returns the previous `enabled`.
 */
int hgr_gemm_set_tail(int enabled, int full_panels);
/*
 * Persistent form of the residual-producer GEMMs (hgr_gemm_nt_res_stats*): on launches of one to two rounds of the chip's workgroup
 * slots every workgroup walks its share of the tiles (tile b, b + slots, ...) instead of ending after one, so that a tile's
 * read-modify-write of the residual pair drains beside the next tile's first operand loads.  Same tiles, same arithmetic: results are
 * bit-identical.  enabled = 1 (default; HGR_DUO_PERSIST) / 0.  Process-wide development knob like hgr_gemm_set_tile (A/B runs, tests).
 * Returns the previous setting.
 */
int hgr_gemm_set_persist(int enabled);

/*
 * Role-split form of the tower GEMMs (round 5 EXPERIMENT; csrc/hgr_gemm_ws.hip): launches of hgr_gemm_nt (16-bit output, epilogues NONE /
 * BIAS / BIAS_RELU), hgr_gemm_nt_ln and hgr_gemm_nt_res_stats* that are made of whole 256 x 128 tiles (M % 256 == 0, N % 128 == 0,
 * K % 128 == 0, K >= 768, at least one tile per CU) run as ONE persistent workgroup per CU of four matrix waves (LDS fragment reads
 * + MFMA only, fragments prefetched a phase ahead) and four helper waves that issue the operand LDS-DMAs and run the epilogue of tile
 * i - nn.Linear's bias, QuickGELU (clip/model.py:162-164), the folded LayerNorm (clip/model.py:153-159), the residual add of
 * clip/model.py:186-187 - under the MFMAs of tile i + 1.  Same tiles, same K order, same epilogue arithmetic as the
 * two-workgroups-per-CU kernel: results are bit-identical (tests/test_gpu_kernels.py::test_gemm_ws_equals_duo).  It does NOT beat
 * that kernel (the epilogue's vector instructions take the same issue slots from the SIMD's matrix wave whichever wave runs them;
 * both forms are bound by LDS bandwidth - DESIGN.md 4.1, profiles/NOTES.md round 5), so enabled = 0 is the default (HGR_WS=1 / this
 * call switch it on: A/B runs, tests).  Process-wide development knob like hgr_gemm_set_tile.  Returns the previous setting.
 */
int hgr_gemm_set_ws(int enabled);

/*
 * The LayerNorm-folded consumer GEMM (hgr_gemm_nt_ln) also exists as ONE persistent 512-thread workgroup per CU on 256 x 256 tiles
 * (csrc/hgr_gemm_p8.hip: the main loop of hgr_gemm_nt_ln_mha with a store epilogue): 2/3 of the staged bytes per flop of the
 * 256 x 128 two-workgroup kernel, bit-identical to it (tests/test_gpu_kernels.py::test_gemm_p8_equals_duo).  It wins with long K and
 * many tiles per CU (25 600 x 3 072 x 3 072: 1 108 -> 1 228 TF/s; ViT-L/14's c_fc -4.4 %) and is level or behind on ViT-B/32's shapes,
 * so mode 2 (the default) picks it by shape: whole 256 x 256 tiles, K >= 1 024, >= 4 tiles per CU.  0 = never, 1 = wherever whole
 * tiles cover the launch (A/B runs, tests).  Process-wide development knob like hgr_gemm_set_tile.  Returns the previous mode.
 */
int hgr_gemm_set_p8(int mode);

/*
 * Patch extraction for the ViT stem: image fp32 NCHW [B,3,R,R] -> rows [B*g*g, Kp] of 16-bit
 * values, g = R/P, row (b, gy, gx) holds the patch in (c, py, px) order = conv1.weight.reshape(W,-1)
 * order, zero-padded from 3*P*P to Kp (Kp % 64 == 0).  With hgr_gemm_nt this replaces
 * `self.conv1(x)` + reshape + permute at clip/model.py:220-222 (kernel = stride = P, no bias).
 */
int hgr_im2col_patches(const float *image, void *out, int B, int R, int P, int Kp, int dtype, void *stream);
/* Same, but patch p of image b goes to output row b*rows_per_image + row_offset + p: with (L, 1) the patches land
 * directly in the [B, L] token layout with the class-token rows left untouched (training keeps them zero). */
int hgr_im2col_patches_ex(const float *image, void *out, int B, int R, int P, int Kp, int rows_per_image, int row_offset,
                          int dtype, void *stream);
/* Input-pipeline form: uint8 NHWC crops [B, R, R, 3] (what PIL / a JPEG decoder hands over) -> normalised 16-bit
 * patch rows in (py, px, c) order, fusing ToTensor + Normalize(mean, std) of clip/clip.py:71-78 into the patch
 * extraction.  mean3 / std3 are HOST pointers to 3 floats.  The conv weight must be permuted to the same K order
 * (weight.permute(0,2,3,1)); a quarter of the bytes of the fp32 path cross PCIe and HBM. */
int hgr_im2col_patches_u8(const unsigned char *image, void *out, int B, int R, int P, int Kp, int rows_per_image,
                          int row_offset, const float *mean3, const float *std3, int dtype, void *stream);
/*
 * Input pipeline on the device: Resize(n_px, BICUBIC) + CenterCrop(n_px) [+ ToTensor + Normalize] of a batch of decoded
 * RGB images - the reference's `_transform` (clip/clip.py:71-78 = dataset/imagenet_group.py:27-34), whose resize runs
 * inside Pillow (libImaging/Resample.c: two-pass separable convolution, 22-bit fixed-point taps, 8-bit clipped
 * intermediate).  Bit-exact: the taps are computed on the host in double precision (hgr_net_amd.preprocess) and the
 * device does the integer convolutions.
 *   src     packed uint8 images; image b = h[b] x w[b] x 3 bytes (row-major HWC, RGB) at src + off[b]
 *   off     int64 [B] (device);  hw int32 [B, 2] = (h, w) (device)
 *   tab     int32 [B] (device) or NULL: index of image b's tap tables (images of one size share them); NULL = b
 *   xb, yb  int32 [T, R, 2] (device): (first source column / row, tap count) of column / row i of the CROPPED output
 *   xk, yk  int32 [T, R, KX] / [T, R, KY] (device): the taps, zero padded to the batch maximum; KX % 4 == 0 and xk
 *           16-byte aligned (a row's bytes are fetched 12 at a time: `src` needs 32 readable bytes past its last image)
 *   out_u8  uint8 [B, R, R, 3] or NULL (feeds hgr_im2col_patches_u8);  out_f32 fp32 [B, 3, R, R] or NULL
 *   mean3, std3  HOST pointers to 3 floats (needed for out_f32)
 */
int hgr_preprocess_bicubic(const unsigned char *src, const int64_t *off, const int *hw, const int *tab, const int *xb, const int *xk, int KX,
                           const int *yb, const int *yk, int KY, unsigned char *out_u8, float *out_f32,
                           const float *mean3, const float *std3, int B, int R, void *stream);

/* In place on the [B*L, W] fp32 patch-GEMM output with zero class rows: t[b,l] += positional[l] + (l == 0 ? class : 0)
 * (clip/model.py:223-224; the un-fused form of hgr_vit_embed_ln, whose pre-LayerNorm sum the backward needs). */
int hgr_vit_assemble(float *t, const float *class_embedding, const float *positional_embedding, int B, int L, int W, void *stream);

/*
 * ViT token assembly + ln_pre (clip/model.py:223-225): x[b,0] = class_embedding, x[b,1+i] = patch
 * row i; + positional_embedding; LayerNorm(eps) -> fp32 residual stream x [B*L, W], L = G + 1.
 *   patches fp32 [B*G, W] (output of the patch GEMM)
 */
int hgr_vit_embed_ln(const float *patches, const float *class_embedding, const float *positional_embedding,
                     const float *gamma, const float *beta, float *x,
                     int B, int G, int W, float eps, void *stream);

/*
 * Row LayerNorm in fp32 (clip/model.py:153-159), fp32 in, 16-bit (`dtype`) or fp32 out.
 * Source row of output row r is  x + (r*row_mul + (row_idx ? row_idx[r] : 0)) * W  so the same call
 * serves ln_1/ln_2 (row_mul 1), ln_post on token 0 (row_mul L; clip/model.py:231) and ln_final on the
 * EOT rows (row_mul L, row_idx = EOT position; clip/model.py:346-350).  W % 4 == 0, W <= 4096.
 */
int hgr_layernorm(const float *x, const float *gamma, const float *beta, void *y,
                  int rows, int W, int64_t row_mul, const int32_t *row_idx,
                  float eps, int dtype, int out_f32, void *stream);

/*
 * Multi-head self-attention core, d_head = 64 (clip/model.py:259,268,417), replacing the
 * softmax(q k^T / sqrt(d) + mask) v part of nn.MultiheadAttention (clip/model.py:171,183 and
 * :324-330 for the causal mask).  qkv 16-bit [B*L, 3W] packed (q | k | v), token rows batch-major
 * (row = b*L + t); out 16-bit [B*L, W].  causal != 0 masks key > query.  1 <= L <= 288.
 */
int hgr_mha(const void *qkv, void *out, int B, int L, int heads, int causal, int dtype, void *stream);
/* The same for the first q_rows query rows of every sequence only (1 <= q_rows <= L): rows q_rows .. L-1 of `out` are left
 * untouched, the rows that are written carry the bits hgr_mha writes.  A ViT's last block needs the class token's row alone
 * (clip/model.py:231 reads x[:, 0, :]): its keys / values are still all L tokens. */
int hgr_mha_rows(const void *qkv, void *out, int B, int L, int heads, int causal, int q_rows, int dtype, void *stream);
/* The same, and per (batch, head, query) the softmax statistics of the row: stats[((b * heads + h) * L + q) * 2] = max of the
 * scaled scores, [.. + 1] = 1 / sum of exp(score - max) (fp32 [B, heads, L, 2], 8-byte aligned).  The training forward keeps
 * them so that hgr_mha_bwd_stats does not recompute Q K^T for the statistics (what autograd's saved softmax output holds). */
int hgr_mha_stats(const void *qkv, void *out, float *stats, int B, int L, int heads, int causal, int dtype, void *stream);

/*
 * Text embedding (clip/model.py:340-342): x[i*L + t] = token_embedding[tokens[i, t]] + positional[t],
 * t < L <= ctx (L < ctx = EOT trimming, exact under the causal mask), fp32 out [n*L, W].
 * tokens int64 [n, ld_tokens].  Token ids are range-checked on device (clamped, error flag not raised).
 */
int hgr_text_embed(const int64_t *tokens, int64_t ld_tokens, const float *token_embedding,
                   const float *positional_embedding, float *x, int n, int L, int W, int vocab, void *stream);

/* eot[i] = argmax_t tokens[i, t] (first maximum), the row picked at clip/model.py:350. int32 out. */
int hgr_eot_index(const int64_t *tokens, int64_t ld_tokens, int32_t *eot, int n, int ctx, void *stream);

/*
 * Row L2 normalisation (model/clip_tree.py:323,330): y = x / ||x||_2 per row, x fp32 [rows, D].
 * Writes y16 (16-bit `dtype`, may be NULL) and/or y32 (fp32, may be NULL, may alias x).
 */
int hgr_l2norm_rows(const float *x, void *y16, float *y32, int rows, int D, int dtype, void *stream);

/*
 * Top-k per row over a column subset, largest first, ties to the lowest subset position
 * (`logits[:, index].topk(k)`: main.py:136-139 with k = 20, :157 with k = 1).
 *   logits fp32 [rows, ld]; cols int32 [n_cols] or NULL (= columns 0..n_cols-1)
 *   out_idx int32 [rows, k] = node ids (cols[pos]), out_val fp32 [rows, k] or NULL
 * 1 <= k <= 32, n_cols <= 40000 (one row is staged in the 160 KB LDS).
 */
int hgr_topk_rows(const float *logits, int64_t ld, const int32_t *cols, int n_cols, int k,
                  int32_t *out_idx, float *out_val, int rows, void *stream);

/*
 * Level-segmented arg-max (the hierarchy "segmented reduce", main.py:162-176): for every row and
 * every depth level l < n_levels, the node id  cols[p*]  where p* is the first position maximising
 * v_p = (depth[cols[p]] == l ? logits[row, cols[p]] : -1)  over the subset positions p, which is what
 * `logits.index_fill(1, rest, -1)[:, train_index].topk(1)` computes.  One coalesced pass over the row.
 *   depth int32 [n_nodes] (shortest-path depth, utils.py:55,66-70); out int32 [rows, n_levels]
 *   out_top1 int32 [rows] or NULL: the unmasked top-1 node id over the same subset
 *   (`logits[:, train_index].topk(1)`, main.py:157), which is the best of the per-level bests.
 * n_levels <= 32; every depth[cols[p]] must be < n_levels.
 */
int hgr_level_argmax(const float *logits, int64_t ld, const int32_t *cols, int n_cols,
                     const int32_t *depth, int n_levels, int32_t *out, int32_t *out_top1, int rows, void *stream);

/*
 * Fused per-row evaluation = hgr_level_argmax + its top-1 + hgr_topk_rows in one kernel (main.py:136-176).  Two fully
 * coalesced sweeps over the logits row (HBM once, L2 once), driven by dense per-column maps built once per model:
 *   lvl8 uint8 [n_nodes] depth of node c;  train_pos / test_pos int32 [n_nodes] = position of column c in the train / test
 *   subset or -1 (positions only break ties, like `logits[:, index].topk`);  train_cols / test_cols int32 = position ->
 *   node id;  filler_pos int32 [n_levels] = first train position NOT at depth l (or -1): the reference's -1 fill competes.
 *   - sweep 1 keeps, per thread, the running best of every depth level in registers (the segments of the reduce are the
 *     levels) plus the thread's best test element, then one shuffle reduction per level; out_level [rows, n_levels],
 *     out_top1 [rows] (may be NULL) = best of the level bests;
 *   - sweep 2 collects the top-k candidates (>= the k-th best slice maximum), ranked by brute force: out_topk [rows, k]
 *     (k = 0 skips it).  Exact for any data, ties to the lowest subset position; n_levels <= 32.
 */
int hgr_eval_rows(const float *logits, int64_t ld, int n_nodes, const unsigned char *lvl8, const int32_t *train_pos,
                  const int32_t *train_cols, int n_train, int n_levels, const int32_t *filler_pos, const int32_t *test_pos,
                  const int32_t *test_cols, int n_test, int k, int32_t *out_level, int32_t *out_top1, int32_t *out_topk,
                  int rows, void *stream);

/* The counters of one evaluation batch (main.py:139-148 top-k hits, :157-160 hit_ratio, :177-191 path / point overlap) from
 * hgr_eval_rows' outputs: acc[0..4] += rows whose target is inside the top-1/2/5/10/20, acc[5] += top-1 hits on any node of
 * the target's path, acc[6] += matched consecutive level pairs / (L - 1) (matches when L == 1), acc[7] += matched levels / L,
 * acc[8] += rows.  targets int64 [rows] or NULL (then `target` for every row); parents / levels int32 [L]: the path's node
 * ids and their depths; acc double [9] on the device, advanced in place (one block, fixed order). */
int hgr_eval_counters(const int32_t *pred, int k, const int64_t *targets, int target, const int32_t *top1, const int32_t *lv,
                      int n_levels, const int32_t *parents, const int32_t *levels, int L, double *acc, int rows, void *stream);

/* ------------------------------------------------------------------------------------------------
 * ModifiedResNet (RN50) tower, clip/model.py:93-150.  Activations are NHWC 16-bit ([B, H, W, C] =
 * a row-major [B*H*W, C] matrix), so every 1x1 convolution IS hgr_gemm_nt; inference BatchNorm
 * (the model is always in eval(): model/clip_tree.py:46) is folded into the weights and a bias.
 * ------------------------------------------------------------------------------------------------ */

/*
 * 3x3 convolution, padding 1, stride 1 or 2, + folded BN + ReLU as an IMPLICIT GEMM
 * (conv2 of every Bottleneck and the stem's conv2/conv3: clip/model.py:20-21,43,106-109).
 *   x   NHWC 16-bit [B, H, W, C], C a multiple of 8 (any of CLIP's RN widths, padded to 8)
 *   w   16-bit [Cout, Kp], K order (ky, kx, c) = weight.permute(0,2,3,1).reshape(Cout, 9C), zero
 *       padded to Kp (multiple of 64);  bias fp32 [Cout];  out NHWC 16-bit [B, Ho, Wo, Cout]
 * No im2col buffer: the LDS-DMA loader gathers the taps, out-of-bounds taps read a zero page.
 */
int hgr_conv3x3_nhwc(const void *x, const void *w, const float *bias, void *out,
                     int B, int H, int W, int C, int Cout, int stride, int Kp, int dtype, void *stream);

/*
 * relu(conv3x3(x, pad 1, stride 1) + bias) followed by AvgPool2d(2) in one kernel: the end of the ModifiedResNet stem
 * (clip/model.py:108-109 `relu(bn3(conv3(x)))`, `avgpool(x)`).  x NHWC 16-bit [B, H, W, 32] (H, W even), w / bias as for
 * hgr_conv3x3_nhwc, out NHWC [B, H/2, W/2, Cout], Cout in {32, 64}.  Same values as the two separate calls (the pool averages
 * the 16-bit conv outputs in fp32); the full-resolution tensor is never written.
 */
int hgr_conv3x3_pool2_nhwc(const void *x, const void *w, const float *bias, void *out, int B, int H, int W, int C, int Cout,
                           int Kp, int dtype, void *stream);

/*
 * The stem's first convolution in one kernel (clip/model.py:106 `relu(bn1(conv1(x)))`, 3 x 3, stride 2, pad 1, 3 -> Cout):
 *   out NHWC 16-bit [B, Ho, Ho, Cout] = relu(conv(image fp32 [B, 3, R, R]) + bias),  Ho = (R - 1) / 2 + 1
 * w 16-bit [Cout, Kp] in (ky, kx, c) order, zero beyond column 27 (the folded layout); R a multiple of 4, Cout a multiple
 * of 8 up to 48.  Replaces hgr_stem_im2col + hgr_gemm_nt on the inference path (no [pixels, 64] matrix in HBM).
 */
int hgr_stem_conv1(const float *image, const void *w, const float *bias, void *out, int B, int R, int Cout, int Kp,
                   int dtype, void *stream);

/*
 * im2col of the stem's first convolution (3 -> width/2 channels, 3x3, stride 2, pad 1,
 * clip/model.py:105) straight from the fp32 NCHW image: out 16-bit [B*Ho*Wo, 64], K order
 * (ky, kx, c), 27 values + zero padding.  Followed by hgr_gemm_nt with HGR_EPI_BIAS_RELU.
 */
int hgr_stem_im2col(const float *image, void *out, int B, int R, int dtype, void *stream);

/* 2x2 average pooling, stride 2 (nn.AvgPool2d(2): clip/model.py:22,34,111), NHWC 16-bit, C % 8 == 0. */
int hgr_avgpool2_nhwc(const void *x, void *out, int B, int H, int W, int C, int dtype, void *stream);

/*
 * AttentionPool2d token assembly (clip/model.py:66-69): tokens[b, 0] = mean over the S*S cells,
 * tokens[b, 1 + i] = cell i; + positional_embedding.  x NHWC 16-bit [B, S, S, C]; pos fp32
 * [S*S + 1, C]; out 16-bit [B*(S*S+1), C].  C % 8 == 0.
 */
int hgr_attnpool_tokens(const void *x, const float *pos, void *out, int B, int S, int C, int dtype, void *stream);

/*
 * AttentionPool2d attention for the ONLY query that is used, token 0 (the module returns x[0],
 * clip/model.py:90): out[b, h*64 + d] = sum_j softmax_j(q[b,h] . k[b,j,h] / 8) v[b,j,h,d].
 *   q fp32 [B, E] (projected mean token), k, v 16-bit [B*L, E], out 16-bit [B, E]; E = heads*64, L <= 256.
 */
int hgr_attnpool_attend(const float *q, const void *k, const void *v, void *out, int B, int L, int heads,
                        int dtype, void *stream);

/* ------------------------------------------------------------------------------------------------
 * OM training step (model/clip_tree.py:222-281 `train_batch`, main.py:86-94): backward + optimizer.
 * GEMM gradients reuse hgr_gemm_nt: dX = dY . W is an NT product against the pre-transposed weight,
 * dW += dY^T . X an NT product of the transposed operands with HGR_EPI_ACCUM.  Parameter gradients are
 * fp32 and ACCUMULATE (autograd `.grad +=` semantics, needed by the K.M inner `loss_j.backward()` calls).
 * ------------------------------------------------------------------------------------------------ */

/*
 * Weight gradients with the operands as they lie in memory ("TN": the reduction index m is the ROW index of both):
 *   partial[s][a][b] = sum over rows m in [s*kc, min(M, (s+1)*kc)) of P[m][a] * Q[m][b]      (fp32, s < ceil(M/kc))
 * For a Linear / 1x1 convolution dW[n][k] = sum_m dY[m][n] X[m][k] (autograd behind clip_tree.py:279-280): P = dY,
 * Q = X, no transposed copies.  P [M, ldp], Q [M, ldq] 16-bit; Na, Nb, ldp, ldq multiples of 8; kc a multiple of 64;
 * partial [S][Na][ldo] (ldo >= Nb, ldo % 4 == 0).  Reduce the S slices with hgr_colsum.
 */
int hgr_gemm_tn_splitk(const void *P, int64_t ldp, const void *Q, int64_t ldq, float *partial, int64_t ldo,
                       int M, int Na, int Nb, int kc, int dtype, void *stream);
/* Output tile edge (256 or 128) hgr_gemm_tn_splitk runs for an [Na, Nb] product: 256 x 256 tiles occupy a CU alone (size
 * kc for ~256 workgroups per round), 128 x 128 tiles run two per CU (~512).  Pure function of the shape (HGR_TN_TILE=128|256
 * in the environment pins it, for A/B runs); not an error code. */
int hgr_gemm_tn_tile(int Na, int Nb);

/* The same for a 3x3 / pad 1 / stride 1 convolution (every 3x3 of clip/model.py's Bottleneck and stem conv2/conv3):
 *   partial[s][co][(ky, kx, c)] = sum over pixels m of slice s of dY[m][co] * x[pixel m shifted by (ky-1, kx-1)][c]
 * x NHWC 16-bit [B, H, W, C] (C a multiple of 8), dy [B*H*W, lddy] with Cout live columns; the loader gathers the taps
 * (zero outside the image) - no im2col buffer.  partial [S][Cout][ldo], ldo >= 9*C. */
int hgr_conv3x3_wgrad_splitk(const void *dy, int64_t lddy, const void *x, float *partial, int64_t ldo,
                             int B, int H, int W, int C, int Cout, int kc, int dtype, void *stream);

/* y[c][r] = x[r][c] for 16-bit elements; ldy >= rows (columns r >= rows of y are left untouched). */
int hgr_transpose16(const void *x, int64_t ldx, void *y, int64_t ldy, int rows, int cols, void *stream);

/* hgr_transpose16 and, from the same pass over x, out[c] = (accumulate ? out[c] : 0) + alpha * sum_r x[r][c]: the
 * transposed dY that `dW += dY^T X` needs and the bias gradient `db += colsum(dY)` (autograd's Linear / Conv2d backward
 * behind clip_tree.py:279-280) in one read.  ldx, ldy multiples of 8, 16-byte aligned bases;
 * scratch >= ceil(rows/64) * cols floats; summation order is fixed. */
int hgr_transpose16_colsum(const void *x, int64_t ldx, void *y, int64_t ldy, int rows, int cols, int dtype,
                           float *out, int accumulate, float alpha, float *scratch, void *stream);

/* out[c] = (accumulate ? out[c] : 0) + alpha * sum_r x[r][c]  (bias gradients, positional-embedding gradients).
 * x is fp32 (x_f32) or 16-bit `dtype`; scratch >= ceil(rows/512) * cols floats; summation order is fixed. */
int hgr_colsum(const void *x, int64_t ldx, int rows, int cols, int x_f32, int dtype, float *out, int accumulate,
               float alpha, float *scratch, void *stream);

/* y = (16-bit) x, n % 4 == 0. */
int hgr_cast16(const float *x, void *y, int64_t n, int dtype, void *stream);
/* y [rows, ldy] = (16-bit) x [rows, ldx] and yt [cols, ldyt] = its transpose, one pass (hgr_cast16's roundings).  The per-step refresh
 * of a Linear's training operands: the forward / weight-gradient operand W and the [K, N] operand of dX = dY W (utils.py:117-123 is
 * the reference's per-step fp32 <-> fp16 weight conversion).  Columns [rows, ldyt) of yt are not written. */
int hgr_cast16_transpose(const float *x, int64_t ldx, void *y, int64_t ldy, void *yt, int64_t ldyt, int rows, int cols, int dtype, void *stream);

/* QuickGELU on 16-bit tensors (clip/model.py:162-164): forward out = a*sigmoid(1.702a); backward out = du * g'(a). */
int hgr_quickgelu16(const void *a, const void *du, void *out, int64_t n, int backward, int dtype, void *stream);

/*
 * LayerNorm backward (autograd of clip/model.py:153-159): dx[src(r)] += dLN(dy[r]); dgamma += sum_r dy*xhat;
 * dbeta += sum_r dy.  dy 16-bit `dtype` or fp32 (dy_f32) [rows, W]; x fp32 (forward input); src(r) as in
 * hgr_layernorm.  scratch >= hgr_layernorm_bwd_scratch_floats(rows, W) floats.  Bit-reproducible.
 */
int hgr_layernorm_bwd(const void *dy, int dy_f32, const float *x, const float *gamma, float *dx, float *dgamma,
                      float *dbeta, float *scratch, int rows, int W, int64_t row_mul, const int32_t *row_idx,
                      float eps, int dtype, void *stream);
/* The same, and the rows it touched of the UPDATED dx also as a 16-bit copy dx16[src row][W] (= hgr_cast16 of those rows):
 * the next GEMMs of the backward chain take the gradient in 16 bit, this saves the separate conversion pass. */
int hgr_layernorm_bwd_cast(const void *dy, int dy_f32, const float *x, const float *gamma, float *dx, void *dx16, float *dgamma,
                           float *dbeta, float *scratch, int rows, int W, int64_t row_mul, const int32_t *row_idx, float eps,
                           int dtype, void *stream);
/* The same, and dx16_colsum[c] += sum over the touched rows of dx16[:, c] AS ROUNDED (fp32 [W], accumulated): dx16 is the output
 * gradient of the Linear that closes the residual branch behind this LayerNorm (out_proj / c_proj, clip/model.py:186-187), so this
 * is that Linear's bias gradient - without the hgr_colsum pass over dx16 it otherwise costs. */
int hgr_layernorm_bwd_cast_colsum(const void *dy, int dy_f32, const float *x, const float *gamma, float *dx, void *dx16, float *dgamma,
                                  float *dbeta, float *dx16_colsum, float *scratch, int rows, int W, int64_t row_mul,
                                  const int32_t *row_idx, float eps, int dtype, void *stream);
int64_t hgr_layernorm_bwd_scratch_floats(int rows, int W);

/* Attention backward for hgr_mha (L <= 320): dqkv [B*L, 3W] from qkv, the forward output `out` and dout [B*L, W], all
 * 16-bit.  Short sequences (L <= 32) use per-output fp32 loops; longer ones a tiled flash-style kernel on the exact-fp32
 * MFMA that recomputes P from row statistics (nothing L x L is stored). */
int hgr_mha_bwd(const void *qkv, const void *out, const void *dout, void *dqkv, int B, int L, int heads, int causal, int dtype, void *stream);
/* The same with the row statistics hgr_mha_stats wrote for this qkv (same B, L, heads, causal): identical results, one sweep
 * over the score blocks less.  (L <= 32 runs the single-block kernel, which does not read them.) */
int hgr_mha_bwd_stats(const void *qkv, const void *out, const void *dout, void *dqkv, const float *stats,
                      int B, int L, int heads, int causal, int dtype, void *stream);
/* The same (stats may be NULL: recomputed), and colsum_part [B, 3W] (fp32, every entry written) = the column sums of sequence b's L
 * rows of dqkv AS ROUNDED.  Summed over b (hgr_colsum) that is the in_proj bias gradient - autograd's grad_output.sum(0) behind
 * clip/model.py:171 - without a second pass over the [B*L, 3W] gradient. */
int hgr_mha_bwd_colsum(const void *qkv, const void *out, const void *dout, void *dqkv, const float *stats, float *colsum_part,
                       int B, int L, int heads, int causal, int dtype, void *stream);

/*
 * ---- ModifiedResNet tower in training (clip/model.py:10-150 under model/clip_tree.py:222-281; the reference's README
 * trains --arch RN50).  BatchNorm uses running statistics even in training (clip_tree.py:46), so every conv + BN pair
 * is one folded convolution; its weight / bias gradients are mapped back to conv.weight, bn.weight, bn.bias. ----
 */
/* 3x3 / pad 1 / stride 1 convolution without bias or activation: the data gradient of hgr_conv3x3_nhwc when `w` holds
 * the spatially flipped, in/out-transposed folded weight [Cin, Kp >= 9*Cout]. */
int hgr_conv3x3_nhwc_plain(const void *x, const void *w, void *out, int B, int H, int W, int C, int Cout, int Kp,
                           int dtype, void *stream);
/* Split-K NT product for weight gradients (tiny output, reduction over all pixels): split s = 0 .. ceil(K/kc)-1 writes
 * partial[s] [M, ldc] fp32 = A[:, s*kc : (s+1)*kc] . W[:, same]^T; the caller sums the partials (hgr_colsum). */
int hgr_gemm_nt_splitk(const void *A, int64_t lda, const void *W, int64_t ldw, float *partial, int64_t ldc,
                       int M, int N, int K, int kc, int dtype, void *stream);
/* out = (y > 0) ? dy : 0 (ReLU backward from the saved output; out may alias dy); n % 8 == 0. */
int hgr_relu_bwd16(const void *dy, const void *y, void *out, int64_t n, int dtype, void *stream);
/* out = a + b, 16-bit with an fp32 add (the two gradient branches of a Bottleneck); n % 8 == 0. */
int hgr_add16(const void *a, const void *b, void *out, int64_t n, int dtype, void *stream);
/* AvgPool2d(2) backward: dx [B, H, W, C] = dy [B, H/2, W/2, C] / 4 per covered pixel. */
int hgr_avgpool2_bwd_nhwc(const void *dy, void *dx, int B, int H, int W, int C, int dtype, void *stream);
/* AttentionPool2d token assembly backward (clip/model.py:67-69): dx [B, S, C] = dtok[:, 1:] + dtok[:, 0] / S. */
int hgr_attnpool_tokens_bwd(const void *dtok, void *dx, int B, int S, int C, int dtype, void *stream);
/* Transposed im2col for the 3x3 weight gradient: xt [C, ld] (activation transposed, column = pixel) ->
 * out [9*C, ld], row (tap, c) = xt row c shifted by the tap, zero outside the image and in the padding columns. */
int hgr_im2col3x3_t(const void *xt, void *out, int B, int H, int W, int C, int64_t ld, int dtype, void *stream);
/* Fold inference BatchNorm into its convolution: w16 [Cout, Kp] in (ky, kx, ci) order = conv.weight * gamma/sigma,
 * bias = beta - mean * gamma/sigma (khw = 1 or 9). */
int hgr_bn_fold(const float *w, const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                void *w16, float *bias, int Cout, int Cin, int khw, int Kp, int dtype, void *stream);
/* Gradients of the folded weight gwf [Cout, ldg] (same K order) and bias gbf [Cout] -> += into the gradients of
 * conv.weight [Cout, Cin, kh, kw], bn.weight and bn.bias. */
int hgr_bn_unfold_grad(const float *gwf, int64_t ldg, const float *gbf, const float *w, const float *gamma, const float *mean,
                       const float *var, float eps, float *g_w, float *g_gamma, float *g_beta, int Cout, int Cin, int khw,
                       void *stream);

/*
 * DGP baseline graph propagation (baseline/DGP/models/gcn_dense_att.py:31-46 `GraphConv.forward` after the dense
 * `support = x W + b`): out[i] = act( sum_e att[grp[e]] * inv_deg[e] * (support[col[e]] + bias) ) over the edges of row
 * i in a CSR that merges the D distance-grouped, in-degree-normalised operators (baseline/DGP/utils.py:56-65;
 * grouping: materials/make_dense_grouped_graph.py:25-38).  act = LeakyReLU(slope) (slope 1 = identity), then an optional
 * row L2 normalisation (`F.normalize`, gcn_dense_att.py:115).
 *   work items  item_row / item_e0 / item_e1 / item_slot int32 [n_items]: edges [e0, e1) of `row`; slot < 0 = the row's
 *               only item (finished in place), else the index of its fp32 partial in `partial` [n_slots, C]
 *   split rows  split_row / split_slot0 / split_n int32 [n_split]: rows cut into several items, summed in slot order
 *   col int32 [nnz], inv_deg fp32 [nnz], grp uint8 [nnz], att fp32 [D] (already soft-maxed), bias fp32 [C] or NULL
 *   support fp32 [n, ld_support], out fp32 [n, ld_out]; C % 4 == 0, C <= 4096; every row of `out` needs an item
 */
int hgr_csr_group_aggregate(const float *support, int64_t ld_support, const int *item_row, const int *item_e0,
                            const int *item_e1, const int *item_slot, int n_items, const int *col, const float *inv_deg,
                            const unsigned char *grp, const float *att, int D, const float *bias,
                            const int *split_row, const int *split_slot0, const int *split_n, int n_split,
                            float *partial, float *out, int64_t ld_out, int C, float slope, int normalize, void *stream);

/*
 * Row-wise softmax cross-entropy (nn.CrossEntropyLoss at clip_tree.py:49,275): loss_rows[r] = lse(logits[r]) -
 * logits[r, labels[r]]; dlogits (optional) = (softmax - onehot) * gscale (gscale = weight / rows for the mean).
 */
int hgr_ce_rows(const float *logits, int64_t ld, const int32_t *labels, int rows, int n, float gscale,
                float *loss_rows, float *dlogits, int64_t ldd, void *stream);

/* Backward of y = x/|x| (clip_tree.py:225,262): dx (+)= (dy - y (y.dy)) / |x|, fp32 rows. */
int hgr_l2norm_bwd(const float *x, const float *dy, float *dx, int rows, int D, int accumulate, void *stream);

/* Small fp32 product with generic strides: C[m][n] (+)= alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]
 * (the [B, <=257] logits head and the two projections: tiny, kept in fp32 for the loss path). */
int hgr_matmul_f32(const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn, float *C,
                   int64_t ldc, int M, int N, int K, float alpha, int accumulate, void *stream);

/* out (+)= alpha * <a, b> over n fp32 elements, fixed summation order (d logit_scale = sum(dlogits * logits), clip_tree.py:263). */
int hgr_dot_f32(const float *a, const float *b, int64_t n, float *out, float alpha, int accumulate, void *stream);

/* token_embedding gradient: dtable[tokens[i,t]] += dx[i*L + t] (fp32 atomics). */
int hgr_embed_scatter_add(const int64_t *tokens, int64_t ld_tokens, const float *dx, float *dtable, int n, int L,
                          int W, int vocab, void *stream);

/* dst[(r*dst_mul + idx[r])] += alpha * src[r]  for W-wide fp32 rows (scatter of EOT / class-token row gradients;
 * dst_mul = 0 with distinct idx: scatter-add of per-step text-feature gradients into the de-duplicated prompt set). */
int hgr_rows_axpy(float *dst, int64_t dst_mul, const int32_t *dst_idx, const float *src, int rows, int W, float alpha, void *stream);

/* CoOp learnable prompts (reference model/CoOp.py:58-113; BASELINE configs[4]): after hgr_text_embed, rows 1..n_ctx of every
 * prompt become the shared context vectors: x[i*L + 1 + c] = ctx[c] + positional[1 + c].  Backward: dctx[c] += sum_i
 * dx[i*L + 1 + c] (fixed order) and those dx rows are cleared, so the placeholder tokens' embeddings get no gradient. */
int hgr_ctx_splice(float *x, const float *ctx, const float *positional_embedding, int n, int L, int W, int n_ctx, void *stream);
int hgr_ctx_splice_bwd(float *dx, float *dctx, int n, int L, int W, int n_ctx, void *stream);

/* dst[r] = src[idx[r]]  for W-wide fp32 rows (the text features of one inner step out of the de-duplicated set). */
int hgr_rows_gather(float *dst, const float *src, const int32_t *idx, int rows, int W, void *stream);

/* *out += sum x^2 (global gradient norm of clip_grad_norm_, main.py:88).  Deterministic (fixed summation order).  The block partials
 * live in process-wide device scratch, handed out as a ring of 16 slots to consecutive calls: calls on ONE stream (the product path)
 * need nothing; more than 16 calls in flight at once on different streams would share a slot and must be ordered by the caller. */
int hgr_sumsq(const float *x, int64_t n, float *out, void *stream);

/*
 * Fused AdamW step with torch.optim.AdamW semantics (main.py:247,91) on fp32 master parameters; if sumsq_total
 * is given, gradients are first scaled by min(1, max_norm / (sqrt(*sumsq_total) + 1e-6)) = clip_grad_norm_.
 * grad_scale multiplies every gradient first (1/world after a sum all-reduce; 1 otherwise).
 */
int hgr_adamw(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
              float wd, int step, const float *sumsq_total, float max_norm, float grad_scale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm folded into the GEMMs around it.  A residual block of the reference is
 *     x = x + attn(ln_1(x));  x = x + c_proj(QuickGELU(c_fc(ln_2(x))))          (clip/model.py:185-188, LayerNorm :153-159)
 * i.e. every LayerNorm sits between a GEMM that PRODUCES its input row (out_proj / c_proj + residual add) and a GEMM that
 * CONSUMES its output (in_proj / c_fc).  With
 *     LN(x) W^T + b = rstd * ( x (gamma o W)^T - mean * s ) + c,    s_n = sum_k gamma_k W_nk,   c_n = sum_k beta_k W_nk + b_n
 * the consumer runs on the un-normalised 16-bit rows with the gamma-folded weight and the producer supplies per-row partial
 * (sum, sum of squares) per 64-column slot: the separate LayerNorm pass (118 MB of traffic per call at ViT-B/32 batch 512)
 * disappears.  Between the GEMMs the residual stream is kept as a PAIR (xh, xl): xh = x rounded to the MFMA type - it IS the
 * consumer's A operand - and xl ONE BYTE q per element (round 4; rounds 2 - 3: xl = f16(x - xh)), both cut out of x's fp32 bit
 * pattern.  With S = 13 (f16) / 16 (bf16) mantissa bits dropped by xh:
 *     t = bits(x) + 2^(S-1);   bits(xh) = t with its low S bits cleared;   q = bits S-1 .. S-8 of t;
 *     decode: bits(x') = bits(xh) + ((q - 128) << (S - 8))  =  bits(x) with its low S - 8 bits cleared
 * i.e. x to 8 more mantissa bits than xh holds - 19 / 16 significant bits in 3 bytes, |x - x'| < ulp(xh) / 256; xh is x rounded to
 * nearest with ties away from zero (round-to-nearest-even except on exact ties); below the f16 normal range the error is absolute
 * (<= 2^-25) (csrc/hgr_common.h: pair_split / pair_dec, used by every kernel that touches the pair; hgr_pair_rows_f32 is the decoder
 * for callers).  Error study of the folding: tools/studies/ln_fusion_study.py.
 *   stats  fp32 [M][N/64][2]  (N = row width);  xh 16-bit [M, ldx], xl uint8 [M, ldx] (same element stride);  requirements: row
 *   width % 128 == 0; xh / xl 16-byte
 *   aligned with ldx % 8 == 0 (the producer reads and writes 8 columns per lane: 16 + 8 bytes) and ldx < 2^23; the consumer's C 16-byte
 *   aligned with ldc % 8 == 0, ldc < 2^23 (a tile's rows are addressed at 32-bit byte offsets from a 64-bit tile base: strided
 *   row views such as the class-token rows of a [B, L, W] stream, row stride L * W, are fine).
 * ------------------------------------------------------------------------------------------------ */
/* (xh, xl) += A W^T + bias  (the residual add of clip/model.py:186-187 on the pair, in place), stats = slot partials of the new rows */
int hgr_gemm_nt_res_stats(const void *A, int64_t lda, const void *W, int64_t ldw, void *xh, void *xl, int64_t ldx,
                          const float *bias, float *stats, int M, int N, int K, int dtype, void *stream);

/*
 * The same with a RANGE GUARD on the residual stream.  The reference keeps LayerNorm in fp32 (clip/model.py:153-159) because trained
 * CLIP residual streams carry outlier channels; here the stream's high half xh is 16-bit, so a value beyond the f16 range would
 * become inf.  Whenever a 64-column slot's sum of squares of the NEW rows exceeds guard_sumsq - or is inf / NaN - its bit pattern
 * is atomicMax'ed into *flag (zeroed by the caller; it stays 0 while every |x| <= sqrt(guard_sumsq)).  Nearly free: the slot sums
 * exist anyway, and no atomic is issued while the stream is in range.  The host reads the flag at its next synchronisation point
 * and falls back to the unfused fp32 stream (hgr_net_amd.clip.model: CLIP.ln_guard_tripped).  flag = NULL: no guard.
 */
int hgr_gemm_nt_res_stats_guard(const void *A, int64_t lda, const void *W, int64_t ldw, void *xh, void *xl, int64_t ldx,
                                const float *bias, float *stats, float guard_sumsq, uint32_t *flag,
                                int M, int N, int K, int dtype, void *stream);

/*
 * Visual head of a ViT in one launch (clip/model.py:231-233: x = ln_post(x[:, 0, :]); x = x @ proj) on the residual stream kept as
 * a pair (above): row b of the result = LayerNorm(decode(xh, xl)[b * row_mul]; gamma, beta, eps) rounded to the MFMA type,
 * times proj_t^T (proj_t [D, W] 16-bit = visual.proj transposed), fp32 [B, D].  row_mul = tokens per image (the class token is
 * token 0).  LayerNorm arithmetic = hgr_layernorm's (two-pass statistics in fp32).  W % 32 == 0, W <= 1920, D % 4 == 0.
 */
int hgr_vit_head(const void *xh, const void *xl, int64_t ldx, int64_t row_mul, const float *gamma, const float *beta, float eps,
                 const void *proj_t, float *out, int B, int W, int D, int dtype, void *stream);
/* C (16-bit) = act( rstd_m (XH Wfold^T - mean_m ln_s) + ln_c ), act: 0 none (ln_1 -> in_proj), 1 QuickGELU (ln_2 -> c_fc -> gelu);
 * K = row width, mean / rstd from `stats` ([M][K/64][2], as written by the producers), eps of the LayerNorm */
int hgr_gemm_nt_ln(const void *XH, int64_t ldx, const void *Wfold, int64_t ldw, void *C, int64_t ldc,
                   const float *ln_s, const float *ln_c, const float *stats, float eps,
                   int M, int N, int K, int dtype, int act, void *stream);

/*
 * The attention half of a residual block in one launch (clip/model.py:171,183-186: `self.attn(x, x, x)` on `self.ln_1(x)` -
 * nn.MultiheadAttention's in_proj, the scaled dot products, softmax and the value product; out_proj stays hgr_gemm_nt_res_stats):
 *     att [B * L, heads * 64] = softmax(q k^T / 8 [+ causal mask]) v per (sequence, head),   [q | k | v] = LN(x) W_in^T + b_in
 * with the LayerNorm folded as in hgr_gemm_nt_ln (XH, Wfold [3 W, W], ln_s / ln_c [3 W], stats, eps as there).  A tile is
 * floor(256 / L) WHOLE sequences x the 192 q / k / v columns of one head, so the [B * L, 3 W] qkv tensor is never written or
 * re-read and no attention launch follows.  Same bits as hgr_gemm_nt_ln into a qkv buffer followed by hgr_mha on it.
 * 1 <= L <= 64 (ViT-B/32: 50, trimmed prompts, the attention pool); row width W = heads * 64, W % 128 == 0.
 */
int hgr_gemm_nt_ln_mha(const void *XH, int64_t ldx, const void *Wfold, int64_t ldw, const float *ln_s, const float *ln_c,
                       const float *stats, float eps, void *att, int64_t ldatt, int B, int L, int heads, int causal,
                       int dtype, void *stream);

/* Training forward of the QuickGELU MLP (clip/model.py:177-180 under autograd): one GEMM, two 16-bit outputs -
 *   pre  [M, ldpre]  = A W^T + bias                (kept for the backward: HGR_EPI_QGELU_GRAD16 / hgr_quickgelu16 read it)
 *   post [M, ldpost] = g(pre as rounded to 16 bit)  (the input of c_proj) - the bits hgr_quickgelu16 would write from `pre`.
 * Same shape contract as hgr_gemm_nt_ln (N % 128 == 0, K >= 128, operands below 4 GB). */
int hgr_gemm_nt_bias_gelu_dual(const void *A, int64_t lda, const void *W, int64_t ldw, void *pre, int64_t ldpre, void *post, int64_t ldpost,
                               const float *bias, int M, int N, int K, int dtype, void *stream);
/* Backward of the same MLP, data gradient of c_proj with the QuickGELU derivative and the bias gradient of c_fc in one launch:
 *   C [M, ldc] (16-bit) = (A W^T) * g'(pre)              - hgr_gemm_nt with HGR_EPI_QGELU_GRAD16, bit for bit
 *   colsum_part [ceil(M / 64), N] (fp32, every entry written) = column sums of C AS ROUNDED over rows [64 u, 64 u + 64)
 * i.e. d(c_fc.bias) = sum_u colsum_part[u] (hgr_colsum over it), without the second pass over the [M, N] gradient that
 * hgr_colsum(C) costs (the reference's autograd sums grad_output per Linear: torch/nn/functional.linear backward behind
 * model/clip_tree.py:274 `loss.backward()`).  Shape contract of hgr_gemm_nt_ln (N % 128 == 0, K >= 128, operands below 4 GB). */
int hgr_gemm_nt_qgelu_grad_colsum(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc, const void *pre, int64_t ldpre,
                                  float *colsum_part, int M, int N, int K, int dtype, void *stream);
/* hgr_vit_embed_ln whose output rows leave as the pair + slot statistics (input of the first block) */
int hgr_vit_embed_ln_stats(const float *patches, const float *class_embedding, const float *positional_embedding,
                           const float *gamma, const float *beta, void *xh, void *xl, float *stats,
                           int B, int G, int W, float eps, int dtype, void *stream);
/* x fp32 [rows, W] -> pair + slot statistics (text tower: embedding rows feed the first block) */
int hgr_row_stats16(const float *x, void *xh, void *xl, float *stats, int rows, int W, int dtype, void *stream);
/* out fp32 [rows, W] (compact) = xh[src] + xl[src], src = i * row_mul + (row_idx ? row_idx[i] : 0): selected rows of the pair back in
 * fp32 (ln_post on the class tokens clip/model.py:233, ln_final on the EOT rows :349-350) */
int hgr_pair_rows_f32(const void *xh, const void *xl, float *out, int rows, int W, int64_t row_mul, const int32_t *row_idx, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------------
 * hgr_logits_eval: the class-logits GEMM of tree_model.forward (model/clip_tree.py:331) with the evaluation consumers of main.test
 * (main.py:136-176: top-20 over the test columns, top-1 over the train columns, arg-max per depth level) in its epilogue, so the
 * [rows, N] logits are never written or re-read.  Same outputs, bit for bit, as hgr_gemm_nt (fp32 logits) followed by
 * hgr_eval_rows.  The caller prepares, once per classifier, a LEVEL-SORTED class matrix: columns ordered by depth level, every
 * level padded with zero rows to a multiple of 32 columns (and the total to a multiple of 96 = one CU's slab), so that each 32-column
 * slice lies inside one level:
 *   zsl_perm16 16-bit [n_perm, D] (D = row stride);  tpos_perm / epos_perm int32 [n_perm]: position of the column in
 *   train_index / test_index or -1 (padding: both -1);  level_first int32 [n_levels + 1]: level l owns the 32-column slices
 *   [level_first[l], level_first[l + 1]) (an empty level: an empty range; trailing padding slices belong to no level);
 *   filler_pos / train_cols / test_cols exactly as for hgr_eval_rows;  feat16: L2-normalised image features, 16-bit [rows, D].
 * Stage 1 (round 4: one 512-row x 96-column tile per CU, the class matrix streams through every CU once): per (row, slice) the
 * best train column as an orderable key and, per 16-column group, the largest test value, its position and the runner-up -> workspace.
 * Stage 2 (one workgroup per row): level arg-max / top-1 from the keys; top-k threshold = k-th largest slice maximum; the few groups
 * that hold two candidates are recomputed on the matrix cores (same operand roles and K order: identical bits) and ranked.
 * workspace: hgr_logits_eval_workspace_bytes(rows, n_perm) bytes, 16-byte aligned.  D % 128 == 0, D <= 1024, n_levels <= 32,
 * n_perm % 96 == 0, n_perm <= 32768.
 * ------------------------------------------------------------------------------------------------ */
int64_t hgr_logits_eval_workspace_bytes(int rows, int n_perm);
int hgr_logits_eval(const void *feat16, const void *zsl_perm16, int rows, int D, int n_perm,
                    const int32_t *tpos_perm, const int32_t *epos_perm, const int32_t *level_first,
                    int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                    const int32_t *test_cols, int n_test, int k,
                    int32_t *out_level, int32_t *out_top1, int32_t *out_topk, void *workspace, int dtype, void *stream);
/* The two stages of hgr_logits_eval on their own (same arguments): measurement only - hgr_logits_eval is exactly the pair, and the row
 * stage reads the workspace a tile stage with the same arguments wrote. */
int hgr_logits_eval_tile_stage(const void *feat16, const void *zsl_perm16, int rows, int D, int n_perm,
                    const int32_t *tpos_perm, const int32_t *epos_perm, const int32_t *level_first,
                    int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                    const int32_t *test_cols, int n_test, int k,
                    int32_t *out_level, int32_t *out_top1, int32_t *out_topk, void *workspace, int dtype, void *stream);
int hgr_logits_eval_row_stage(const void *feat16, const void *zsl_perm16, int rows, int D, int n_perm,
                    const int32_t *tpos_perm, const int32_t *epos_perm, const int32_t *level_first,
                    int n_levels, const int32_t *filler_pos, const int32_t *train_cols, int n_train,
                    const int32_t *test_cols, int n_test, int k,
                    int32_t *out_level, int32_t *out_top1, int32_t *out_topk, void *workspace, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel collectives over RCCL / xGMI (one process per GPU).  The reference has no distributed code (its only
 * mention is an unused DDP import, baseline/CLIP/clip_train.py:19); these serve the sharding this build adds:
 *   evaluation  - every rank text-encodes N/world prompts of update_classifier (model/clip_tree.py:318-325) and the
 *                 row slices are ALL-GATHERED; the 9 metric counters of main.test (main.py:121-128) are ALL-REDUCED once;
 *   OM training - one single-class batch is sharded over the ranks, the flat fp32 gradient buffer is ALL-REDUCED (sum)
 *                 before clip_grad_norm_ + AdamW (main.py:87-91); parameters can be BROADCAST from a root.
 * Bootstrap: rank 0 calls hgr_comm_unique_id and ships the HGR_COMM_ID_BYTES bytes to every rank by any side channel
 * (file, TCP store, torch.distributed); every rank then calls hgr_comm_init with its HIP device current.  Calls are
 * stream-ordered on `stream` and asynchronous like every other entry point.  librccl.so.1 is resolved at run time
 * (HGR_RCCL_LIB overrides the path), so the library loads without it; the calls then fail with HGR_EUNSUPPORTED.
 * ------------------------------------------------------------------------------------------------ */
#define HGR_COMM_ID_BYTES 128
typedef enum { HGR_COMM_F32 = 0, HGR_COMM_F64 = 1, HGR_COMM_F16 = 2, HGR_COMM_BF16 = 3, HGR_COMM_I32 = 4, HGR_COMM_I64 = 5, HGR_COMM_U8 = 6 } hgr_comm_dtype_t;
typedef enum { HGR_COMM_SUM = 0, HGR_COMM_MAX = 1 } hgr_comm_op_t;

int hgr_comm_unique_id(void *id_out /* HGR_COMM_ID_BYTES */);
int hgr_comm_init(int rank, int world, const void *unique_id /* HGR_COMM_ID_BYTES */);
int hgr_comm_destroy(void);
int hgr_comm_rank(void);    /* -1 without a communicator */
int hgr_comm_world(void);   /*  0 without a communicator */
/* recv[i] = op over ranks of send[i]; in place (send == recv) allowed */
int hgr_allreduce(const void *send, void *recv, int64_t count, int dtype, int op, void *stream);
/* recv[r * count_per_rank + i] = rank r's send[i]; in place when send == recv + rank * count_per_rank */
int hgr_allgather(const void *send, void *recv, int64_t count_per_rank, int dtype, void *stream);
int hgr_broadcast(void *buf, int64_t count, int dtype, int root, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HGR_H */
