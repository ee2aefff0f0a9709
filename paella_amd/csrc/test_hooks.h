/* Test / tooling hooks exported by libpaella_hip.so but NOT part of the public C ABI (include/paella_hip.h).
 * Bound by paella_amd/_lib.py (TEST_HOOKS) for tests/ and tools/ only.  The switches are process-wide plain variables read at launch time:
 * set them from one host thread, while no other thread is enqueueing work (none of them is reachable from the product path). */
#ifndef PAELLA_TEST_HOOKS_H
#define PAELLA_TEST_HOOKS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* one GEMM on bf16 operands (the opt-in fast mode's kernels: bit patterns A16 [M,K], W16 [N,K] supplied by the caller) with an explicit tile config
 * (10, 18, 19, 30..37; < 0 = heuristic) / workgroup count as paella_op_gemm; ln_stats != NULL: LayerNorm of the A rows folded into the epilogue from
 * [M, K/16, 2] (sum, centred M2) partials; C16 != NULL: also store the result rounded to bf16 (C may then be NULL) */
int paella_test_gemm_bf16(const unsigned short* A16, const unsigned short* W16, const float* bias, const float* residual, float* C, unsigned short* C16,
                          int M, int N, int K, int act, const float* ln_stats, int tile_cfg, int splitk, void* ws, size_t ws_bytes, void* stream);
/* the LayerNorm-folding bf16 GEMM as the model launches it: A16 = the bf16 copy of the fp32 rows A32, ln_stats = [M, K/16, 2] partials of the fp32 rows; blocks whose
 * |mean| / std exceeds the fold threshold (paella_test_ln_fold_ratio) re-read A32, normalise in fp32 and round the normalised operand to bf16 */
int paella_test_gemm_bf16_ln(const unsigned short* A16, const float* A32, const unsigned short* W16, float* C, int M, int N, int K, const float* ln_stats,
                             int tile_cfg, int splitk, void* ws, size_t ws_bytes, void* stream);
/* the bf16 attention core of the opt-in fast mode (>= 256 queries in the model): q16 / ks16 / vs16 bf16 [B*L, nhead*D], conditioning k / v fp32, out16 bf16 */
int paella_test_attention_bf16(const unsigned short* q16, const unsigned short* ks16, const unsigned short* vs16, const float* k_cond, const float* v_cond,
                               unsigned short* out16, int B, int nhead, int D, int Lq, int Lself, int Lcond, const float* key_weights, int n_kw, void* stream);
/* the fast mode's GlobalResponseNorm apply, in place on a bf16 tensor [rows, C]: h = bf16(h * scale[row / rows_per_sample][c] + shift[c]) (fp32 arithmetic, one rounding) */
int paella_test_grn_apply16(unsigned short* h, const float* scale, const float* shift, int64_t rows, int rows_per_sample, int C, void* stream);
/* A/B of the bf16 tile rules: bit 0 = never the 256x128 / 256x256 tiles (the fp32 rules' tiles instead), bit 1 = no persistent ranges of the 256x128 tile,
 * bit 2 = never the 256x256 ping-pong tile (long-K launches take the 256x128 tile as in round 5) */
int paella_test_gemm_bf16_rule(int mask);
/* launches n_launches dependent, nearly empty kernels (blocks x 256 threads touching n_elems floats): boundary floor */
int paella_test_launch_chain(float* buf, int n_elems, int blocks, int n_launches, void* stream);
/* 1 = run large-query-count attention on the register-fed kernel instead of the LDS-staged one; 10 = LDS-staged kernel with register staging (rounds 2-5),
 * 11 = its padded direct-to-LDS layout at every head_dim; 0 = default = direct-to-LDS staging, unpadded (4 workgroups per CU) at odd head_dim / 16, padded otherwise
 * (attention.hip: STG; bit-identical outputs; A/B probe tools/attn_probe.py) */
int paella_test_attention_variant(int v);
/* C = prologue(A) . W^T with an explicit tile config / workgroup count (as paella_op_gemm): mode 1: a' = a * scale[row / rows_per_sample][k] +
 * shift[k] (the GRN apply of the MLP's second GEMM); mode 2: a' = (a - mean) * rstd from ln_stats [M, K/16, 2] = per 16-column block (sum, M2 =
 * sum of squared deviations from the block mean) (LayerNorm folded into the consumer's EPILOGUE; the hook sums W's rows itself with one extra M = 1 launch per call) */
int paella_test_gemm_prologue(const float* A, const float* W, float* C, int M, int N, int K, int mode, const float* scale, const float* shift,
                              int rows_per_sample, const float* ln_stats, int tile_cfg, int splitk, void* ws, size_t ws_bytes, void* stream);
/* out[M, c] = GRN(gelu(h W1^T + b1)) W2^T with GlobalResponseNorm finished inside the two GEMMs (the batch-1 path of a ResBlock's MLP: no finalize
 * launch); scratch: hidden [M, 4c], gx [M / rps, 4c], part [M / rps, 4c / 16].  Fails when the shape is outside the fused path's domain. */
int paella_test_mlp_grn_fused(const float* h, const float* W1, const float* b1, const float* gamma, const float* beta, const float* W2, float* hidden,
                              float* gx, float* part, float* out, int M, int c, int rps, void* ws, size_t ws_bytes, void* stream);
/* 0 = never use the direct-to-LDS (DMA) twins of the large GEMM tiles (A/B and parity checks); 1 = default */
int paella_test_gemm_dma(int on);
/* the LDS-DMA ring tile (config id 30..35) the launch heuristic uses for the skinny batch-1 shapes; 0 = the register-staged / 1-deep kernels (A/B) */
int paella_test_gemm_ring(int cfg);
/* per-launch-site workgroup count of the skinny (ring-tile) GEMM class: site = (M, N, K, prologue class 0 / 1 GRN / 2 LayerNorm, bf16 operands 0 / 1); G > 0 sets it,
 * G == 0 removes the site's run-time entry, M == 0 removes all run-time entries (tools/site_tune.py) */
int paella_test_gemm_site(int M, int N, int K, int apro, int bf, int G);
/* the same with an explicit tile id for sites of ANY class (cfg < 0: workgroup count only) */
int paella_test_gemm_site_cfg(int M, int N, int K, int apro, int bf, int cfg, int G);
/* the launch heuristic's table of resident workgroups (whole chip) of ring tile cfg (30..35) with operand prologue class apro (0 none, 1 GRN, 2 LayerNorm); -1 otherwise */
long paella_test_ring_resident(int cfg, int apro);
/* tile of the fused head GEMM + sampling tail: 9 = 128x128, 14 = 128x64 on 8 waves (several workgroups per CU), 18 = 64x64 direct-to-LDS (four workgroups per CU; default) */
int paella_test_gemm_tail_tile(int cfg);
/* tile rows per rasterisation group of the GEMM (default 8); 0 = plain m-fastest tile order (A/B) */
int paella_test_gemm_raster(int gm);
/* 256x128 tile (bf16 operands only): which of the two waves that share a SIMD runs its LDS-DMA issue / fragment reads late (under the other one's MFMA block): 0 = neither
 * (both straight after the barrier), 1 = waves 4..7 (default), 2 = odd waves */
int paella_test_gemm_big_stagger(int mode);
/* |mean| / std above which a 16-row block of a LayerNorm-consuming GEMM normalises its operand fragments instead of folding the LayerNorm into the
 * epilogue (default 4; inf = always fold, 0 = never): measures the fold's error curve (tests/test_gpu_ops.py, profiles/r04_ln_fold_error_curve.txt) */
int paella_test_ln_fold_ratio(float ratio);
/* dev_word != NULL: every LayerNorm-consuming GEMM launch adds to *dev_word the number of its waves whose 16-row blocks took the operand-side LayerNorm (the
 * guard tripped); NULL (default) = off.  Lets a whole-network test assert that the guard really ran inside the model (tests/test_gpu_unet.py) */
int paella_test_ln_guard_counter(unsigned* dev_word);
/* 0 = GlobalResponseNorm always through the grn_from_partials finalize launch (A/B of the in-GEMM statistics of the batch-1 path); 1 = default */
int paella_test_grn_fuse(int on);
/* Measurement hook (bench.py roofline line): when enabled, EVERY dense-contraction launch (the head GEMM with the fused sampling tail included)
 * is bracketed by HIP events on its stream; collect() returns the summed duration, algorithmic FLOPs and bytes since enable(1).  Process-wide,
 * single host thread (the events live in a plain vector). */
int paella_prof_enable(int on);
int paella_prof_collect(double* total_ms, double* total_flops, double* total_bytes, int64_t* launches);
/* per-launch records since enable(1), not reset (call before collect): us_out[i], shape_out[5 i ..] = M, N, K, prologue (0 none, 1 GRN, 2 LayerNorm,
 * 3 implicit convolution), fused-tail flag; returns the launch count (at most cap are written) */
long long paella_prof_detail(float* us_out, int* shape_out, long long cap);
/* scores_out [rows, L] = the Gumbel-max scores of the counter-based sampling tail (mix(l_c, l_u) / T - log q, the kernels' own arithmetic and
 * Philox counters): tests classify a differing token by the decision margin between the two best scores of its row */
int paella_test_tail_scores(const float* logits_c, const float* logits_u, int64_t rows, int L, float cfg, float one_minus_cfg, float temperature,
                            uint64_t seed, uint64_t offset, int64_t row_offset, float* scores_out, void* stream);
#ifdef __cplusplus
}
#endif
#endif
