// Host-side AddressSanitizer driver of the C-ABI shim (SURVEY.md section 5 "Race detection / sanitizers").
// Built and run by `make -C hgr-net_amd/csrc asan` against libhgr_asan.so = the same sources with the HOST code compiled
// under -fsanitize=address (device code is not instrumented: GPU ASan / xnack is not available on this pool, and no kernel
// is launched here).  Every call below must be rejected by the host-side argument validation BEFORE any launch - null
// operands, bad shapes, misaligned pointers - return a negative code and leave a message in hgr_last_error(); ASan watches
// the validation code, the error formatting (vsnprintf into the thread-local buffer) and the knob setters.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/hgr.h"

static int failures = 0, calls = 0;
#define EXPECT_FAIL(expr) do { ++calls; int rc__ = (expr); const char *m__ = hgr_last_error(); \
    if (rc__ >= 0 || !m__ || !strlen(m__)) { printf("NOT REJECTED (%d): %s\n", rc__, #expr); ++failures; } } while (0)
#define EXPECT_OK(expr) do { ++calls; long rc__ = (long)(expr); if (rc__ < 0) { printf("FAILED (%ld): %s: %s\n", rc__, #expr, hgr_last_error()); ++failures; } } while (0)

int main() {
    // host buffers stand in for device pointers: nothing is dereferenced on the host and nothing is launched
    static float f32[4096] __attribute__((aligned(64)));
    static unsigned short h16[4096] __attribute__((aligned(64)));
    static int i32[256] __attribute__((aligned(64)));
    static long long i64[256] __attribute__((aligned(64)));
    static unsigned char u8[256] __attribute__((aligned(64)));
    char *mis = (char *)h16 + 2;                      // 2-byte aligned only
    EXPECT_OK(hgr_abi_version() == HGR_ABI_VERSION ? 0 : -1);
    // GEMM family
    EXPECT_FAIL(hgr_gemm_nt(nullptr, 64, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_NONE, 1, nullptr));
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 65, HGR_F16, HGR_EPI_NONE, 1, nullptr));      // K % 64
    EXPECT_FAIL(hgr_gemm_nt(h16, 60, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_NONE, 1, nullptr));      // lda < K
    EXPECT_FAIL(hgr_gemm_nt(mis, 64, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_NONE, 1, nullptr));      // misaligned A
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 2, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_NONE, 1, nullptr));       // ldc < N
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 64, 7, HGR_EPI_NONE, 1, nullptr));            // dtype
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_BIAS, 1, nullptr));      // bias missing
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 64, f32, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_BIAS_RESIDUAL, 1, nullptr)); // residual missing
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_ACCUM, 0, nullptr));     // ACCUM into 16-bit
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 64, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, 99, 1, nullptr));
    EXPECT_FAIL(hgr_gemm_set_tile(77));
    EXPECT_OK(hgr_gemm_set_tile(2)); EXPECT_OK(hgr_gemm_set_tile(0));
    EXPECT_FAIL(hgr_gemm_nt_splitk(h16, 64, h16, 64, f32, 64, 4, 4, 64, 32, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_gemm_tn_splitk(nullptr, 64, h16, 64, f32, 64, 128, 64, 64, 64, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_gemm_nt_res_stats(h16, 128, h16, 128, h16, h16, 100, f32, f32, 4, 100, 128, HGR_F16, nullptr));              // N % 128
    EXPECT_FAIL(hgr_gemm_nt_res_stats(h16, 128, h16, 128, h16, h16, 128, nullptr, f32, 4, 128, 128, HGR_F16, nullptr));          // bias
    EXPECT_FAIL(hgr_gemm_nt_res_stats(h16, 128, h16, 128, h16, h16, 132, f32, f32, 4, 128, 128, HGR_F16, nullptr));              // ldx % 8 (16-byte pair accesses)
    EXPECT_FAIL(hgr_gemm_nt_res_stats(h16, 128, h16, 128, (char *)h16 + 8, h16, 128, f32, f32, 4, 128, 128, HGR_F16, nullptr));  // xh only 8-byte aligned
    EXPECT_FAIL(hgr_gemm_set_tail(2, -1)); EXPECT_FAIL(hgr_gemm_set_tail(1, -5));
    EXPECT_OK(hgr_gemm_set_tail(0, -1)); EXPECT_OK(hgr_gemm_set_tail(1, 3)); EXPECT_OK(hgr_gemm_set_tail(1, -1));
    EXPECT_FAIL(hgr_gemm_set_persist(2)); EXPECT_OK(hgr_gemm_set_persist(0)); EXPECT_OK(hgr_gemm_set_persist(1));
    EXPECT_FAIL(hgr_gemm_nt_res_stats_guard(h16, 128, h16, 128, h16, h16, 128, f32, f32, 0.f, (uint32_t *)i32, 4, 128, 128, HGR_F16, nullptr));   // flag without a guard value
    EXPECT_FAIL(hgr_gemm_nt_res_stats_guard(h16, 128, h16, 128, h16, h16, 128, f32, f32, 1.f, (uint32_t *)(u8 + 2), 4, 128, 128, HGR_F16, nullptr)); // misaligned flag
    EXPECT_FAIL(hgr_vit_head(h16, h16, 64, 50, f32, f32, 1e-5f, h16, f32, 4, 100, 64, HGR_F16, nullptr));                        // W % 32
    EXPECT_FAIL(hgr_vit_head(h16, h16, 64, 50, f32, f32, 1e-5f, h16, f32, 4, 2048, 64, HGR_F16, nullptr));                       // W > 1920
    EXPECT_FAIL(hgr_vit_head(h16, nullptr, 64, 50, f32, f32, 1e-5f, h16, f32, 4, 64, 64, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_mha_rows(h16, h16, 1, 50, 12, 0, 0, HGR_F16, nullptr));                                                       // q_rows < 1
    EXPECT_FAIL(hgr_mha_rows(h16, h16, 1, 50, 12, 0, 51, HGR_F16, nullptr));                                                      // q_rows > L
    EXPECT_FAIL(hgr_pair_rows_f32(h16, nullptr, f32, 4, 64, 1, nullptr, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_gemm_nt_ln(h16, 192, h16, 192, h16, 128, f32, f32, f32, 1e-5f, 4, 128, 192, HGR_F16, 0, nullptr));           // K % 128
    EXPECT_FAIL(hgr_gemm_nt_ln(h16, 128, h16, 128, h16, 128, f32, f32, f32, 1e-5f, 4, 128, 128, HGR_F16, 5, nullptr));           // act
    EXPECT_FAIL(hgr_row_stats16(f32, h16, h16, f32, 4, 100, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_vit_embed_ln_stats(f32, f32, f32, f32, f32, h16, nullptr, f32, 1, 4, 64, 1e-5f, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_logits_eval(h16, h16, 4, 100, 128, i32, i32, i32, 4, i32, i32, 8, i32, 32, 20, i32, i32, i32, f32, HGR_F16, nullptr));       // D % 128
    EXPECT_FAIL(hgr_logits_eval(h16, h16, 4, 128, 128, i32, i32, i32, 4, i32, i32, 8, i32, 32, 20, i32, i32, i32, f32, HGR_F16, nullptr));       // n_perm % 96
    EXPECT_FAIL(hgr_logits_eval(h16, h16, 4, 128, 288, i32, i32, i32, 4, i32, i32, 8, i32, 8, 20, i32, i32, i32, f32, HGR_F16, nullptr));        // n_test < k
    EXPECT_OK(hgr_logits_eval_workspace_bytes(512, 22080) == 512ll * 690 * 32 ? 0 : -1);
    // streaming / normalisation kernels
    EXPECT_FAIL(hgr_layernorm(f32, f32, f32, h16, 4, 6, 1, nullptr, 1e-5f, HGR_F16, 0, nullptr));                                // W % 4
    EXPECT_FAIL(hgr_layernorm(f32, f32, f32, h16, 0, 64, 1, nullptr, 1e-5f, HGR_F16, 0, nullptr));
    EXPECT_FAIL(hgr_layernorm(f32, nullptr, f32, h16, 4, 64, 1, nullptr, 1e-5f, HGR_F16, 0, nullptr));
    EXPECT_FAIL(hgr_l2norm_rows(f32, nullptr, nullptr, 4, 64, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_vit_embed_ln(f32, f32, f32, f32, f32, nullptr, 1, 4, 64, 1e-5f, nullptr));
    EXPECT_FAIL(hgr_mha(nullptr, h16, 1, 50, 12, 0, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_mha(h16, h16, 1, 100000, 12, 0, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_im2col_patches(f32, h16, 1, 224, 31, 3072, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_text_embed(nullptr, 77, f32, f32, f32, 4, 8, 64, 100, nullptr));
    EXPECT_FAIL(hgr_eot_index(nullptr, 77, i32, 4, 77, nullptr));
    // evaluation kernels
    EXPECT_FAIL(hgr_topk_rows(f32, 64, nullptr, 64, 0, i32, nullptr, 4, nullptr));
    EXPECT_FAIL(hgr_topk_rows(nullptr, 64, nullptr, 64, 5, i32, nullptr, 4, nullptr));
    EXPECT_FAIL(hgr_level_argmax(f32, 64, nullptr, 64, nullptr, 4, i32, nullptr, 4, nullptr));
    EXPECT_FAIL(hgr_eval_rows(f32, 64, 64, u8, i32, i32, 8, 99, i32, nullptr, nullptr, 0, 20, i32, i32, i32, 4, nullptr));       // levels > 32
    EXPECT_FAIL(hgr_eval_counters(nullptr, 20, nullptr, 0, i32, i32, 4, i32, i32, 3, (double *)f32, 4, nullptr));
    // ModifiedResNet
    EXPECT_FAIL(hgr_conv3x3_nhwc(h16, h16, f32, h16, 1, 8, 8, 7, 64, 1, 64, HGR_F16, nullptr));                                 // C % 8
    EXPECT_FAIL(hgr_conv3x3_nhwc(h16, h16, f32, h16, 1, 8, 8, 8, 64, 3, 128, HGR_F16, nullptr));                                // stride
    EXPECT_FAIL(hgr_avgpool2_nhwc(nullptr, h16, 1, 8, 8, 8, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_attnpool_attend(nullptr, h16, h16, h16, 1, 50, 32, HGR_F16, nullptr));
    // training helpers
    EXPECT_FAIL(hgr_transpose16(nullptr, 8, h16, 8, 4, 4, nullptr));
    EXPECT_FAIL(hgr_cast16(nullptr, h16, 16, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_matmul_f32(nullptr, 4, 1, f32, 4, 1, f32, 4, 4, 4, 4, 1.f, 0, nullptr));
    EXPECT_FAIL(hgr_ce_rows(nullptr, 8, i32, 4, 8, 1.f, f32, nullptr, 0, nullptr));
    EXPECT_FAIL(hgr_sumsq(nullptr, 16, f32, nullptr));
    EXPECT_FAIL(hgr_adamw(nullptr, f32, f32, f32, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, nullptr, 1.f, 1.f, nullptr));
    EXPECT_OK(hgr_layernorm_bwd_scratch_floats(1000, 768));
    // this round's training entries
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, h16, 64, nullptr, nullptr, 0, 4, 4, 64, HGR_F16, HGR_EPI_QGELU_GRAD16, 0, nullptr));   // pre-activation missing
    EXPECT_FAIL(hgr_gemm_nt(h16, 64, h16, 64, f32, 64, nullptr, h16, 64, 4, 4, 64, HGR_F16, HGR_EPI_QGELU_GRAD16, 1, nullptr));       // fp32 output
    EXPECT_FAIL(hgr_gemm_nt_bias_gelu_dual(h16, 128, h16, 128, h16, 100, h16, 128, f32, 4, 100, 128, HGR_F16, nullptr));               // N % 128
    EXPECT_FAIL(hgr_gemm_nt_bias_gelu_dual(h16, 128, h16, 128, h16, 128, nullptr, 128, f32, 4, 128, 128, HGR_F16, nullptr));           // post missing
    EXPECT_FAIL(hgr_gemm_nt_qgelu_grad_colsum(h16, 128, h16, 128, h16, 100, h16, 128, f32, 4, 100, 128, HGR_F16, nullptr));            // N % 128
    EXPECT_FAIL(hgr_gemm_nt_qgelu_grad_colsum(h16, 128, h16, 128, h16, 128, h16, 128, nullptr, 4, 128, 128, HGR_F16, nullptr));        // colsum_part missing
    EXPECT_FAIL(hgr_mha_stats(h16, h16, nullptr, 1, 50, 12, 0, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_mha_bwd_stats(h16, h16, h16, h16, nullptr, 1, 50, 12, 0, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_mha_bwd_colsum(h16, h16, h16, h16, f32, nullptr, 1, 50, 12, 0, HGR_F16, nullptr));                                  // colsum_part missing
    EXPECT_FAIL(hgr_mha_bwd_stats(h16, h16, h16, h16, f32, 1, 400, 12, 0, HGR_F16, nullptr));                                           // L > 320
    EXPECT_FAIL(hgr_cast16_transpose(f32, 64, h16, 32, h16, 64, 64, 64, HGR_F16, nullptr));                                             // ldy < cols
    EXPECT_FAIL(hgr_layernorm_bwd_cast(h16, 0, f32, f32, f32, nullptr, f32, f32, f32, 4, 64, 1, nullptr, 1e-5f, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_layernorm_bwd_cast_colsum(h16, 0, f32, f32, f32, h16, f32, f32, nullptr, f32, 4, 64, 1, nullptr, 1e-5f, HGR_F16, nullptr));
    EXPECT_FAIL(hgr_layernorm_bwd_cast(h16, 0, f32, f32, f32, h16, f32, f32, f32, 4, 6, 1, nullptr, 1e-5f, HGR_F16, nullptr));         // W % 4
    EXPECT_OK(hgr_gemm_tn_tile(3072, 768) == 256 && hgr_gemm_tn_tile(200, 4096) == 128 && hgr_gemm_tn_tile(640, 640) == 128 ? 0 : -1);
    // collectives without a communicator / with bad arguments (librccl may be absent: both outcomes are failures by contract)
    EXPECT_FAIL(hgr_allreduce(f32, f32, 16, HGR_COMM_F32, HGR_COMM_SUM, nullptr));
    EXPECT_FAIL(hgr_allgather(f32, f32, 16, 99, nullptr));
    EXPECT_FAIL(hgr_broadcast(nullptr, 16, HGR_COMM_F32, 0, nullptr));
    EXPECT_FAIL(hgr_comm_init(3, 2, u8));
    EXPECT_FAIL(hgr_comm_init(0, 1, nullptr));
    EXPECT_OK(hgr_comm_destroy());
    EXPECT_OK(hgr_comm_world() == 0 && hgr_comm_rank() == -1 ? 0 : -1);
    (void)i64;
    printf("abi_asan_driver: %d calls, %d failures\n", calls, failures);
    return failures ? 1 : 0;
}
