/*
 * paella_hip.h -- C ABI of libpaella_hip.so: the MI355X (gfx950) implementation of Paella's sampling hot path.
 *
 * The reference (dome272/Paella) is pure Python on PyTorch and has no FFI of its own; the drop-in
 * boundary is therefore the Python surface sample() / Paella / VQModel (paella_amd/ mirrors it), and this
 * header is what that Python surface -- or any other host language -- binds underneath.  Every entry point
 * names the reference code it replaces (file:line under the reference tree).
 *
 * Conventions
 *   - every pointer named dev_* / documented "device" is a HIP device pointer to contiguous fp32 / int64 data;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued on it and
 *     no entry point on the per-step path synchronises the host;
 *   - functions return 0 on success, a negative PAELLA_ERR_* code otherwise; paella_last_error() returns a
 *     thread-local description of the last failure.  Nothing throws across this boundary;
 *   - the library owns only its repacked weight copies; activations, conditioning caches and workspaces are
 *     caller-owned device buffers whose sizes come from the *_bytes() queries;
 *   - every workspace (`ws`) begins with a small header of split-K arrival tickets: call paella_workspace_init() once
 *     on a freshly allocated workspace (a stream-ordered memset); the kernels leave the header zero afterwards.  One
 *     workspace must not be shared by calls that can run concurrently (one per stream / per captured graph);
 *   - handles are not thread-safe: one host thread per model per GPU (the reference's one-process-per-GPU
 *     layout, src_distributed/train.py:186-189).
 *   - activations inside the library are NHWC; logits are returned position-major [B, H, W, num_labels]
 *     (the Python shell hands the reference's [B, num_labels, H, W] shape back as a permuted view).
 */
#ifndef PAELLA_HIP_H
#define PAELLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAELLA_ABI_VERSION 5

#define PAELLA_OK 0
#define PAELLA_ERR_ARG -1       /* invalid argument / unsupported shape */
#define PAELLA_ERR_HIP -2       /* a HIP runtime call or kernel launch failed */
#define PAELLA_ERR_WORKSPACE -3 /* caller-provided buffer too small */
#define PAELLA_ERR_STATE -4     /* model not finalized / tensor missing */

#define PAELLA_MAX_LEVELS 8
#define PAELLA_MAX_BLOCK_TYPES 8

int paella_abi_version(void);
const char* paella_last_error(void);
/* Hash of the kernel sources this library was built from (paella_amd/_stamp.py); the Python binding refuses a library whose stamp differs
 * from the sources next to it. */
const char* paella_source_stamp(void);

/* Zeroes the header (paella_workspace_header_bytes() bytes) of a freshly allocated workspace; required once before the
 * workspace is first passed to any entry point below.  Enqueued on `stream`, no host synchronisation. */
int paella_workspace_init(void* ws, size_t ws_bytes, void* stream);
size_t paella_workspace_header_bytes(void);

/* ------------------------------------------------------------------------------------------------
 * Denoising UNet ("Paella", reference src/modules.py:109-283; alias DenoiseUNet)
 * ---------------------------------------------------------------------------------------------- */
typedef struct paella_unet paella_unet;

/* Mirrors the constructor arguments of reference src/modules.py:110-112. */
typedef struct paella_unet_config {
    int32_t c_in, c_out, num_labels, c_r, patch_size, c_cond;
    int32_t n_levels;
    int32_t c_hidden[PAELLA_MAX_LEVELS];
    int32_t nhead[PAELLA_MAX_LEVELS];
    int32_t blocks[PAELLA_MAX_LEVELS];
    char level_config[PAELLA_MAX_LEVELS][PAELLA_MAX_BLOCK_TYPES]; /* NUL-terminated, letters C T A F */
    int32_t clip_embd, byt5_embd, clip_seq_len, kernel_size, self_attn;
} paella_unet_config;

int paella_unet_create(const paella_unet_config* cfg, paella_unet** out);
void paella_unet_destroy(paella_unet* m);

/* Load one parameter by its reference state-dict key (e.g. "down_blocks.1.3.attention.attn.in_proj_weight")
 * in the reference's own layout; dev_src is a device pointer.  The library copies and repacks it into
 * kernel layout (NHWC-friendly conv weights, depth-to-space row order, ...).  May be called again later to
 * refresh a tensor. */
int paella_unet_load_tensor(paella_unet* m, const char* key, const float* dev_src, const int64_t* shape, int ndim,
                            void* stream);
/* Host table of the c_r/2 sinusoid frequencies exp(-k*log(1e4)/(c_r/2-1)) exactly as the reference computes
 * them (src/modules.py:215-216).  Optional: if never called the library computes them with expf(). */
int paella_unet_set_timestep_freqs(paella_unet* m, const float* host_freqs, int n);
/* Checks that every tensor the configuration needs has been loaded and builds the execution plan. */
int paella_unet_finalize(paella_unet* m, void* stream);

/* Sizes of the caller-owned buffers for a batch of B samples on an H x W token grid with S conditioning rows
 * per sample (S = S_byt5 + clip_seq_len*[clip] + clip_seq_len*n_clip_image). */
size_t paella_unet_cond_bytes(const paella_unet* m, int B, int S);
size_t paella_unet_workspace_bytes(const paella_unet* m, int B, int H, int W, int S);

/* Step-invariant conditioning work, hoisted out of the sampling loop: gen_c_embeddings
 * (src/modules.py:223-232; list-valued clip_image as utils/modules.py:229-235) followed, per AttnBlock, by
 * kv_mapper (src/modules.py:72-75,77) and the K/V in-projection of those rows (nn.MultiheadAttention).
 * byt5 [B, S_byt5, byt5_embd] (S_byt5 may be 0), clip [B, clip_embd] or NULL, clip_image: n_clip_image
 * pointers to [B, clip_embd].  Result goes to cond_out (paella_unet_cond_bytes(B, S) bytes). */
int paella_unet_cond_prepare(paella_unet* m, const float* byt5, int S_byt5, const float* clip,
                             const float* const* clip_image, int n_clip_image, int B, void* cond_out,
                             size_t cond_bytes, void* ws, size_t ws_bytes, void* stream);

/* gen_c_embeddings alone (src/modules.py:223-232): c_embed_out fp32 [B, S, c_cond]; ws needs
 * paella_unet_workspace_bytes() bytes. */
int paella_unet_c_embeddings(paella_unet* m, const float* byt5, int S_byt5, const float* clip,
                             const float* const* clip_image, int n_clip_image, int B, float* c_embed_out, void* ws,
                             size_t ws_bytes, void* stream);
/* gen_r_embedding (src/modules.py:212-221): r fp32 [B] -> r_embed_out fp32 [B, c_r] */
int paella_unet_r_embedding(paella_unet* m, const float* r, int B, float max_positions, float* r_embed_out,
                            void* stream);

/* OPT-IN fast mode of ONE model, outside the fp32 parity contract (no process-wide state): mode 1 routes the forward's dense contractions whose K is a
 * multiple of 64 through bf16-operand MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulation): bf16 shadow weights (made here / refreshed by finalize), bf16
 * activations between producer and consumer GEMMs (the 4c-wide MLP hidden tensor, LayerNorm and attention outputs, a bf16 copy of the residual stream where
 * a LayerNorm-folding GEMM reads it); the residual stream, statistics, attention, logits and the sampling tail stay fp32.  Mode 0 (default) is the exact
 * fp32 path, bit for bit.  Size workspaces (paella_unet_workspace_bytes) AFTER switching: mode 1 needs room for the bf16 activations.  The shadows stay
 * allocated until paella_unet_destroy (a HIP graph captured in mode 1 never dangles) and the switch to mode 0 does not synchronise. */
int paella_unet_set_precision(paella_unet* m, int mode, void* stream);
int paella_unet_get_precision(const paella_unet* m);

/* One denoising evaluation = Paella.forward (src/modules.py:263-275) with the conditioning already prepared.
 * tokens int64 [B,H,W]; r fp32 [B]; attn_weights (utils/alter_attention.py:23-34) fp32 [n_attn_weights] or NULL;
 * logits_out fp32 [B,H,W,num_labels]. */
int paella_unet_forward(paella_unet* m, const int64_t* tokens, const float* r, const void* cond, int B, int H, int W,
                        int S, const float* attn_weights, int n_attn_weights, float* logits_out, void* ws,
                        size_t ws_bytes, void* stream);
/* The same evaluation when batch rows b, b + n_unique, ... share tokens and r (classifier-free guidance: the conditional
 * and unconditional passes of src/utils.py:44-46 batched as 2 x n_unique rows against a B-row conditioning cache).
 * tokens is int64 [n_unique,H,W] and r fp32 [n_unique]: only the distinct rows are passed.  The blocks ahead of the first attention
 * block never see the conditioning and are computed once for the n_unique distinct rows.  n_unique must divide B;
 * n_unique == B with mix_c == mix_u == 0 is paella_unet_forward.
 * (mix_c, mix_u) != (0, 0) (needs B == 2 * n_unique) additionally folds the guidance mix of src/utils.py:47 through the
 * bias-free linear head (out_mapper, src/modules.py:184-187): logits_out then holds only the n_unique rows
 * mix_c * logits(cond) + mix_u * logits(uncond), fp32 [n_unique,H,W,num_labels], equal to mixing the two outputs up to
 * fp32 rounding. */
int paella_unet_forward_shared(paella_unet* m, const int64_t* tokens, const float* r, const void* cond, int B, int n_unique,
                               float mix_c, float mix_u, int H, int W, int S, const float* attn_weights,
                               int n_attn_weights, float* logits_out, void* ws, size_t ws_bytes, void* stream);

/* One whole sampling step in the counter-based (Philox) noise mode: Paella.forward followed by the sampling tail
 * (src/utils.py:43-54) with out_mapper (src/modules.py:184-187) and the tail FUSED: the categorical / argmax decision is taken
 * on the head GEMM's accumulators, the [rows, num_labels] logits tensor the reference materialises (:44-47) is never written.
 * Arguments as paella_unet_forward_shared + paella_sample_tail_ex; with a guidance mix (B == 2 * n_unique) tokens_out holds
 * n_unique x H x W tokens, without one B x H x W (n_unique == B).  Tokens are bit-identical to forward_shared + sample_tail_ex
 * on the same seed / offset / row_offset. */
int paella_unet_forward_sample(paella_unet* m, const int64_t* tokens, const float* r, const void* cond, int B, int n_unique,
                               float mix_c, float mix_u, int H, int W, int S, const float* attn_weights, int n_attn_weights,
                               float temperature, int mode, uint64_t seed, const uint64_t* seed_ptr, uint64_t offset,
                               int64_t row_offset, const int64_t* row_offset_ptr, const int64_t* init_noise, float t_next,
                               int64_t* tokens_out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sampling tail and add_noise (reference src/utils.py:45-54; src/modules.py:277-283)
 * ---------------------------------------------------------------------------------------------- */
/* logits_c / logits_u: fp32 [rows, L] (logits_u NULL = no classifier-free guidance);
 * mixes l = l_c*cfg + l_u*one_minus_cfg, divides by temperature, draws token = argmax softmax(l)/q with
 * q ~ Exp(1) (== torch.multinomial(softmax, 1)), then optionally renoises against init_noise with
 * u <= t_next.  mode 1 = argmax of the mixed logits (the T=0 extension).
 * noise_q [rows, L] / mask_u [rows]: caller-provided noise for bit-parity with torch; NULL = in-kernel
 * Philox4x32-10 keyed by (seed, offset).  sampled_out (optional) receives the pre-renoise draw. */
int paella_sample_tail(const float* logits_c, const float* logits_u, int64_t rows, int L, float cfg,
                       float one_minus_cfg, float temperature, int mode, const float* noise_q, uint64_t seed,
                       uint64_t offset, const int64_t* init_noise, const float* mask_u, float t_next,
                       int64_t* tokens_out, int64_t* sampled_out, void* stream);

/* Same, with (a) an optional DEVICE-resident seed word added to `seed` (seed_ptr may be NULL): a HIP graph that captured
 * the sampling loop can then be replayed with fresh noise by rewriting that one word; (b) row_offset: the Philox counters
 * are keyed by (row + row_offset), so a batch shard that owns global rows [lo, hi) passes lo * H * W and draws exactly the
 * noise the unsharded call draws for those rows (sharded == unsharded); (c) row_offset_ptr (may be NULL): a
 * DEVICE-resident word added to row_offset, so ONE captured graph serves any batch shard by rewriting that word. */
int paella_sample_tail_ex(const float* logits_c, const float* logits_u, int64_t rows, int L, float cfg,
                          float one_minus_cfg, float temperature, int mode, const float* noise_q, uint64_t seed,
                          const uint64_t* seed_ptr, uint64_t offset, int64_t row_offset, const int64_t* row_offset_ptr,
                          const int64_t* init_noise, const float* mask_u, float t_next, int64_t* tokens_out,
                          int64_t* sampled_out, void* stream);

/* Start tokens of the counter-based noise mode (the reference draws torch.randint(0, num_labels, (B,H,W)) from the global
 * generator, src/utils.py:37 -- a stream that cannot be sharded): tokens_out[i] = Philox(seed + *seed_ptr, i + row_offset +
 * *row_offset_ptr) mod num_labels for i in [0, n).  A shard that owns global rows [lo, hi) passes row_offset = lo * H * W and
 * n = (hi - lo) * H * W and obtains exactly its slice of the unsharded draw.  Either pointer may be NULL. */
int paella_start_tokens(uint64_t seed, const uint64_t* seed_ptr, int64_t row_offset, const int64_t* row_offset_ptr,
                        int num_labels, int64_t n, int64_t* tokens_out, void* stream);

/* x, random_x, mask int64 [B, per_sample]; t fp32 [B].  mask_in NULL -> mask = (u <= t[b]) with u = rand_u
 * (caller noise, [B, per_sample]) or Philox; random_x NULL -> Philox randint(0, num_labels). */
int paella_add_noise(const int64_t* x, const float* t, const int64_t* mask_in, const int64_t* random_x,
                     const float* rand_u, uint64_t seed, uint64_t offset, int num_labels, int B,
                     int64_t per_sample, int64_t* x_out, int64_t* mask_out, void* stream);

/* Token select on the [B,H,W] grid: out[i] = keep(i) ? a[i] : (b ? b[i] : fill) with keep(i) = (mask == NULL || mask[i] != 0) && (flag == NULL ||
 * *flag == 1.0f).  `flag` is a DEVICE fp32 word.  Replaces the two integer elementwise expressions of the eval path that used to run through ATen: the
 * inpainting wrapper's `out * mask + tokens * (1 - mask)` (the recipe src/modules.py:277-283 + src_distributed/utils.py:97-109, extension keep_known) and
 * the batch-sharded sampler's "-1 when the conditioning broadcast was flagged invalid" (paella_amd/dist.py); no host synchronisation, graph-capturable. */
int paella_select_tokens(const int64_t* a, const int64_t* b, const int64_t* mask, const float* flag, int64_t fill, int64_t n,
                         int64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VQGAN (reference src/vqgan.py:45-107)
 * ---------------------------------------------------------------------------------------------- */
typedef struct paella_vqgan paella_vqgan;
typedef struct paella_vqgan_config {
    int32_t levels, bottleneck_blocks, c_hidden, c_latent, codebook_size;
    float scale_factor;
} paella_vqgan_config;

int paella_vqgan_create(const paella_vqgan_config* cfg, paella_vqgan** out);
void paella_vqgan_destroy(paella_vqgan* v);
int paella_vqgan_load_tensor(paella_vqgan* v, const char* key, const float* dev_src, const int64_t* shape, int ndim,
                             void* stream);
int paella_vqgan_finalize(paella_vqgan* v, void* stream); /* synchronises the stream once (reads the BatchNorm statistics / ResBlock gammas to the host) */
/* OPT-IN fast mode of ONE VQGAN (outside the fp32 parity contract, as paella_unet_set_precision): mode 1 runs the MLP of every ResBlock whose width is a
 * multiple of 64 on bf16-operand MFMA with fp32 accumulation (bf16 shadow weights, bf16 LayerNorm output and hidden tensor); everything else stays fp32.
 * Mode 0 (default) is the exact path.  Size workspaces (paella_vqgan_workspace_bytes) AFTER switching. */
int paella_vqgan_set_precision(paella_vqgan* v, int mode, void* stream); /* mode 1 on a finalized model converts the shadows and waits for them; mode 0 frees nothing and never synchronises */
/* h, w = latent grid; covers decode and encode of the matching image size */
size_t paella_vqgan_workspace_bytes(const paella_vqgan* v, int B, int h, int w);
/* decode_indices (src/vqgan.py:103-107): idx int64 [B,h,w] -> image fp32 NCHW [B,3,f*h,f*w], f = 2^levels */
int paella_vqgan_decode_indices(paella_vqgan* v, const int64_t* idx, int B, int h, int w, float* img_out, void* ws,
                                size_t ws_bytes, void* stream);
/* decode (src/vqgan.py:97-101): latents fp32 NCHW [B,c_latent,h,w] (already divided by scale_factor, as encode returns) */
int paella_vqgan_decode(paella_vqgan* v, const float* latents, int B, int h, int w, float* img_out, void* ws,
                        size_t ws_bytes, void* stream);
/* encode (src/vqgan.py:91-95): image fp32 NCHW [B,3,Hp,Wp] -> qe_out, x_out fp32 NCHW [B,c_latent,h,w] (both /scale_factor),
 * idx_out int64 [B,h,w], loss_out fp32 [1] = vq_loss + 0.25*commit_loss.  Any output pointer may be NULL. */
int paella_vqgan_encode(paella_vqgan* v, const float* img, int B, int Hp, int Wp, float* qe_out, float* x_out,
                        int64_t* idx_out, float* loss_out, void* ws, size_t ws_bytes, void* stream);
/* VectorQuantize.forward on rows [rows, c_latent] (src/vqgan.py:94; src_distributed/train.py:156): nearest codebook row per
 * input row -> idx_out int64 [rows], qe_out fp32 [rows, c_latent] (optional), mse_out fp32 [1] = mean((qe - x)^2) (optional;
 * the stand-in's vq_loss == commit_loss). */
int paella_vqgan_quantize_rows(paella_vqgan* v, const float* x, int64_t rows, int64_t* idx_out, float* qe_out,
                               float* mse_out, void* stream);
/* VectorQuantize.idx2vq (src/vqgan.py:104): out fp32 [rows, c_latent] = codebook[idx] */
int paella_vqgan_lookup_rows(paella_vqgan* v, const int64_t* idx, int64_t rows, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Single-op entry points (used by the parity tests and the kernel micro-benchmarks)
 * ---------------------------------------------------------------------------------------------- */
/* C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual); act: 0 none, 1 GELU(erf).  tile_cfg < 0 = heuristic; otherwise a
 * tile id with splitk > 0 -> tiles * splitk workgroups (classic split-K), splitk < 0 -> exactly -splitk workgroups walking
 * balanced contiguous (tile, K-step) ranges.  ws = an initialised workspace (split-K tickets + slabs) or NULL. */
int paella_op_gemm(const float* A, const float* W, const float* bias, const float* residual, float* C, int M, int N,
                   int K, int act, int tile_cfg, int splitk, void* ws, size_t ws_bytes, void* stream);
int paella_op_layernorm(const float* x, float* y, int64_t rows, int C, float eps, void* stream);
/* depthwise 3x3 (+ optional skip concat) + LayerNorm on NHWC x [B,H,W,C]; w/bias in reference layout are NOT
 * accepted here: w is [9][C] ([2][9][C] with skip) */
int paella_op_dwconv_ln(const float* x, const float* skip, const float* w, const float* bias, float* y, int B, int H,
                        int W, int C, float eps, void* stream);
int paella_op_grn_scale(const float* g, const float* gamma, float* scale, float* tmp, int B, int rows_per_sample,
                        int C, void* stream);
/* q [B*Lq, nhead*D]; k/v self [B*Lself, nhead*D]; k/v cond [B*Lcond, nhead*D]; out [B*Lq, nhead*D] */
int paella_op_attention(const float* q, const float* k_self, const float* v_self, const float* k_cond,
                        const float* v_cond, float* out, int B, int nhead, int D, int Lq, int Lself, int Lcond,
                        const float* key_weights, int n_kw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PAELLA_HIP_H */
