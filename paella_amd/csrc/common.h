// Shared device/host declarations for the Paella gfx950 kernels.
// Everything here is fp32 / int64, NHWC ("position-major") activations:
// an activation is a row-major matrix [rows = B*h*w, channels].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PAELLA_OK 0
#define PAELLA_ERR_ARG -1
#define PAELLA_ERR_HIP -2
#define PAELLA_ERR_WORKSPACE -3
#define PAELLA_ERR_STATE -4

void paella_set_error(const char* fmt, ...);

#define HIP_CHECK_RET(expr)                                                          \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            paella_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,           \
                             hipGetErrorString(_e));                                 \
            return PAELLA_ERR_HIP;                                                   \
        }                                                                            \
    } while (0)

#define LAUNCH_CHECK_RET()                                                           \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            paella_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__,       \
                             hipGetErrorString(_e));                                 \
            return PAELLA_ERR_HIP;                                                   \
        }                                                                            \
    } while (0)

// Division of n < 2^31 by a launch constant d as one multiply-high and one shift (host: fast_div_of): the hardware has no integer divide, and a division by a
// runtime value costs ~25 dependent instructions through the float reciprocal -- two of them sat in front of every workgroup's FIRST operand fetch
// (tile = unit / KT, tile_m = tile % tiles_m), which at batch 1 is latency on every one of ~1 000 launches per image.
// d >= 2: l = ceil(log2 d), mul = ceil(2^(31 + l) / d) < 2^32, n / d = (n * mul) >> (31 + l) exactly for every n < 2^31; d == 1: mul = 0, n passes through.
struct FastDiv { unsigned mul, shr, pass; };  // pass = 0xffffffff for d == 1 (mul = 0): branch-free identity
__host__ __device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__umulhi(n, f.mul) + (n & f.pass)) >> f.shr;
#else
    return ((unsigned)(((unsigned long long)n * f.mul) >> 32) + (n & f.pass)) >> f.shr;
#endif
}
static inline FastDiv fast_div_of(unsigned d) {
    FastDiv f = {0u, 0u, 0xffffffffu};
    if (d <= 1) return f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    f.shr = l - 1;
    f.pass = 0u;
    return f;
}

// ---------------------------------------------------------------------------
// GEMM epilogue description (shared by the GEMM kernel and the split-K reducer)
// ---------------------------------------------------------------------------
enum { STORE_PLAIN = 0, STORE_D2S = 1, STORE_PIXSHUF_NCHW = 2 };
enum { ACT_NONE = 0, ACT_GELU = 1 };

struct Epilogue {
    const float* bias;      // [N] or null
    int act;                // ACT_*
    float alpha;            // v *= alpha after the activation (VQGAN gamma[5]); 1 otherwise
    const float* residual;  // [M, ldr] or null; added after alpha
    int ldr;
    const float* ts;        // or null: v = v*(1+ts[b*ts_stride + n]) + ts[b*ts_stride + N + n]
    int ts_stride;          // floats between consecutive samples in ts
    int rows_per_sample;    // rows of this matrix per batch sample (for ts)
    FastDiv rps_div;        // division by rows_per_sample (filled by the launchers)
    int store_mode;         // STORE_*
    int sH, sW, sC;         // STORE_D2S / PIXSHUF: source grid (rows m=(b,y,x)), channels per segment
    int py, px;             // STORE_D2S: extra output offset (transposed-conv phase)
    int n_seg_x;            // STORE_D2S: segments along n are (dy,dx) with dx in [0,n_seg_x); 2 for k2s2, 1 for a phase
    float* rowstat_out;     // optional [M, N/16, 2]: per row and 16-column block, (sum, centred sum of squares M2) of the stored values (LayerNorm-on-load; gemm_device.h)
    float* sumsq_out;       // optional [ceil(M/16), N]: per 16-row group, column sums of the stored values squared (GRN)
    // optional (ring tiles whose rows cover whole samples: grn_rps == 16 or == the tile height): GlobalResponseNorm's Gx finished IN this epilogue --
    // grn_gx_out [samples, N] = sqrt(sum over the sample's rows of value^2) and grn_part_out [samples, grn_np] = per (column tile, wave column) sums of
    // Gx, so the consuming GEMM derives mean_k Gx from grn_np numbers instead of waiting for a finalize launch between the two MLP GEMMs
    float* grn_gx_out;
    float* grn_part_out;
    int grn_rps, grn_np;
    int remap_in, remap_out, remap_off;  // STORE_PLAIN, remap_in > 0: out row = (m/remap_in)*remap_out + m%remap_in + remap_off
    // optional (STORE_PLAIN only): the stored values rounded to bf16 (RNE) at the same [row, ldc] positions -- the A operand of a consuming bf16 GEMM (opt-in fast
    // mode).  With GemmArgs::C == nullptr only this copy is written (the 4c-wide hidden tensor of an MLP block never exists in fp32).
    unsigned short* c16;
};

static inline Epilogue make_epilogue() {
    Epilogue e;
    e.bias = nullptr; e.act = ACT_NONE; e.alpha = 1.f; e.residual = nullptr; e.ldr = 0;
    e.ts = nullptr; e.ts_stride = 0; e.rows_per_sample = 1; e.rps_div.mul = 0; e.rps_div.shr = 0; e.rps_div.pass = 0xffffffffu; e.store_mode = STORE_PLAIN;
    e.sH = e.sW = e.sC = 0; e.py = e.px = 0; e.n_seg_x = 2; e.remap_in = e.remap_out = e.remap_off = 0; e.sumsq_out = nullptr; e.rowstat_out = nullptr;
    e.grn_gx_out = nullptr; e.grn_part_out = nullptr; e.grn_rps = 0; e.grn_np = 0; e.c16 = nullptr;
    return e;
}

// Fused sampling tail of the HEAD GEMM (TAIL instantiations of gemm_nt_kernel; reference src/utils.py:47-50 in the counter-based noise
// mode): instead of storing the [M, N = num_labels] logits, every column tile leaves per row its best (score, label); a tiny second
// kernel (tail.hip: tail_finalize_kernel) picks the winner across tiles and renoises.  score = logit / T - log q (Gumbel-max, Philox
// keyed by (seed, global row, label quad, offset)) or the logit itself (mode 1).
struct FusedTail {
    float temperature;
    int mode;                    // 0 categorical, 1 argmax
    uint64_t seed;
    const uint64_t* seed_ptr;    // optional device-resident seed word added to seed
    uint64_t offset;
    int64_t row_offset;
    const int64_t* row_offset_ptr;  // optional device-resident word added to row_offset (a captured graph replayed for another batch shard)
    float* part_score;           // [M, tiles_n]
    int* part_idx;               // [M, tiles_n]
};

// Implicit-GEMM convolution (VQGAN k4 s2 p1 Conv2d / the 4 output phases of the k4 s2 p1 ConvTranspose2d, reference src/vqgan.py:59-61,
// 81-85): the A operand is never materialised.  Row m = output position (b, yo, xo) on a [Ho, Wo] grid, K index = tap * C + c,
// A[m][tap*C + c] = x[b][yo*stride + oy(tap)][xo*stride + ox(tap)][c] (0 outside the [Hi, Wi] input grid), x = GemmArgs::A in NHWC.
// The gather happens in the GEMM's operand load (per-row offsets + a per-K-step tap offset); needs C % (K step) == 0.
struct ConvGather {
    int enabled;
    int Hi, Wi, C, Ho, Wo, stride, ntaps;
    int tw_log2, oy0, ox0, tsign;  // tap t = (ty, tx) = (t >> tw_log2, t & (2^tw_log2 - 1)); offset (oy, ox) = (oy0 + tsign*ty, ox0 + tsign*tx)
};

struct GemmArgs {
    const float* A; int lda;   // [M, K] row-major
    const float* W; int ldw;   // [N, K] row-major (torch Linear layout)
    float* C; int ldc;         // [M, N]
    int M, N, K;
    // optional A-operand prologue (GlobalResponseNorm apply): a' = a*scale[b][k] + shift[k]
    const float* a_scale;      // [samples, K] or null
    const float* a_shift;      // [K]
    int a_rows_per_sample;
    FastDiv a_rps_div;         // division by a_rows_per_sample (filled by the launchers)
    // or (ring tiles only) the same apply from the producer's UNFINISHED statistics: a' = a * (1 + gamma[k] * gx[b][k] / (mean_k gx[b][:] + 1e-6)) + shift[k],
    // mean from grn_part [samples, grn_np] (Epilogue::grn_part_out of the GEMM that produced A); a_rows_per_sample as above
    const float* grn_gx;       // [samples, K] or null
    const float* grn_gamma;    // [K]
    const float* grn_part;     // [samples, grn_np]
    int grn_np;
    int force_ring_cfg;        // > 0: the launch heuristic uses this ring tile for a skinny problem (the producer of grn_gx must cover whole samples)
    // or (exclusive with a_scale) LayerNorm of the A rows from producer statistics: a' = (a - mean[m]) * rstd[m],
    // mean/var combined (parallel-variance formula, fp64) from ln_stats [M, ln_nblk, 2] = (sum, centred M2) per block (Epilogue::rowstat_out of the GEMM that produced A), K == 16*ln_nblk
    const float* ln_stats;
    int ln_nblk;
    float ln_eps;
    const float* ln_row;       // optional [M, 4]: the rows' FINISHED statistics (mean, rstd, mean - (float)mean, |mean| * rstd) from launch_ln_rowstat_finalize -- set by the launcher in the throughput regime
    float ln_fold_ratio;       // 16-row blocks with |mean| * rstd above this normalise their operand fragments instead of using the fold (set by launch_gemm_cfg)
    const float* ln_wsum;      // [N]: sum_k W[n][k] -- the LayerNorm is folded into the epilogue as rstd * (acc - mean * wsum[n]) (gemm.hip: ln_row_stats)
    unsigned* ln_guard_count;  // test hook (null in the product): += 1 per wave whose rows took the operand-side LayerNorm instead of the fold (paella_test_ln_guard_counter)
    Epilogue ep;
    FusedTail ft;              // used by launch_gemm_tail only
    ConvGather cv;             // cv.enabled: A is an NHWC image gathered on the fly (lda unused, K == ntaps * C)
    // OPT-IN bf16 fast mode (outside the fp32 parity contract): when BOTH are set the contraction runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation --
    // same [M, K] / [N, K] row-major layouts and leading dimensions (in elements) as A / W, both operands straight from HBM to LDS by LDS-DMA.  Needs
    // K % 64 == 0, lda % 8 == 0, ldw % 8 == 0, no GRN prologue, no implicit convolution.  A / W are then unused (W may stay set for bookkeeping).
    const unsigned short* A16;
    const unsigned short* W16;
};

// Launchers (each returns PAELLA_OK or an error code; all work is enqueued on `stream`).
// `ws` is a split-K region: kGemmTicketBytes of arrival tickets (zero when first handed to the library -- paella_workspace_init
// -- and left zero by every launch) followed by slab space for partial tiles; ws_bytes covers both.  ws == nullptr forbids
// any K split.  One region must not be used by two launches that can run concurrently.
struct TailArgs;
static const size_t kGemmMaxTickets = (size_t)1 << 16;                       // one ticket per output tile
static const size_t kGemmTicketBytes = kGemmMaxTickets * sizeof(unsigned);  // 256 KiB header
int launch_gemm(const GemmArgs& g, void* ws, size_t ws_bytes, hipStream_t stream);
// Forces a tile config / workgroup count (autotuner and tests). cfg < 0 -> heuristic.  splitk > 0: tiles * splitk workgroups
// (classic split-K); splitk < 0: exactly -splitk workgroups (balanced contiguous unit ranges).
int launch_gemm_cfg(const GemmArgs& g, int cfg, int splitk, void* ws, size_t ws_bytes, hipStream_t stream);
int gemm_num_tile_configs();
// ring tile the first MLP GEMM must run on so that its epilogue finishes GlobalResponseNorm's Gx (Epilogue::grn_gx_out) and the second one can apply it
// straight from those statistics (GemmArgs::grn_gx): 0 = not applicable (use the grn_from_partials finalize launch)
int gemm_grn_fused_tile(int M, int C4, int C, int rows_per_sample, bool allow_64);
// Head GEMM with the fused tail epilogue: one whole tile per workgroup; g.ft.part_* are [M, gemm_tail_tiles_n(M, N)].
int gemm_tail_tiles_n(int M, int N, bool bf16_operands = false);
int launch_gemm_tail(const GemmArgs& g, hipStream_t stream);
// the tile config launch_gemm_tail uses (the unfused head GEMM is launched with the same one: identical logits bit for bit)
int gemm_tail_config(int M, int N, bool bf16_operands = false);
int launch_tail_finalize(const TailArgs& a, const float* part_score, const int* part_idx, int tiles_n, hipStream_t stream);
// true when a GEMM with these shapes can take the bf16-operand kernels (GemmArgs::A16 / W16)
static inline bool gemm_bf16_ok(int K, int lda, int ldw) { return K > 0 && (K & 63) == 0 && (lda & 7) == 0 && (ldw & 7) == 0; }

// ln_stats [M, nblk, 2] (per 16-column block (sum, centred M2)) -> out4 [M, 4] = (mean, rstd, mean - (float)mean, |mean| * rstd)
int launch_ln_rowstat_finalize(const float* stats, int nblk, int K, float eps, float* out4, int64_t M, const float* A32, unsigned short* A16, int lda, float ratio,
                               unsigned* guard_count, hipStream_t stream);  // A16 != null: rows above `ratio` are rewritten as bf16(LayerNorm(fp32 row)) (bf16 fast mode)

// LayerNorm over the channel dimension of [rows, C]; no learned affine (eps 1e-6),
// optional scalar affine y = ln(x)*(1+g0)+g1 (VQGAN), optional space-to-depth gather:
// s2d != 0: output row (b,y',x') segment (dy,dx) <- input row (b,2y'+dy,2x'+dx); out is [rows/4, 4C].
int launch_layernorm(const float* x, float* y, int64_t rows, int C, float eps, float g_mul, float g_add,
                     int s2d, int H, int W, hipStream_t stream);
// the same with an optional bf16 copy of the output (y16; y may then be null) -- A operand of a bf16 GEMM in the opt-in fast mode
int launch_layernorm16(const float* x, float* y, unsigned short* y16, int64_t rows, int C, float eps, float g_mul, float g_add,
                       int s2d, int H, int W, hipStream_t stream);
// bf16 helpers of the opt-in fast mode (elementwise.hip): GRN apply in place on the bf16 hidden tensor, weight shadow copies, row sums of a bf16 matrix
int launch_grn_apply16(unsigned short* h, const float* scale, const float* shift, int64_t rows, int rows_per_sample, int C, hipStream_t stream);
int launch_grn_partials_apply16(const float* part, const float* gamma, const float* shift, float* scale, unsigned short* h, int B, int rows_per_sample, int C, hipStream_t stream);
int launch_f32_to_bf16(const float* src, unsigned short* dst, size_t n, hipStream_t stream);
int launch_rowsum_bf16(const unsigned short* W, float* out, int N, int K, hipStream_t stream);

// UNet ResBlock front half: depthwise 3x3 (zero pad) + bias, then LayerNorm over channels.
// skip != null: grouped 2C->C variant over cat([x, skip]) (reference src/modules.py:46,57).
int launch_dwconv_ln(const float* x, const float* skip, const float* w, const float* bias, float* y,
                     int B, int H, int W, int C, float eps, hipStream_t stream, unsigned short* y16 = nullptr);  // y16: optional bf16 copy (y may then be null)
// VQGAN ResBlock depthwise half: y = x + (dw3x3_replicate(xt) + bias) * gamma2.
int launch_dwconv_res(const float* x, const float* xt, const float* w, const float* bias, float* y,
                      int B, int H, int W, int C, float gamma2, hipStream_t stream);

// GlobalResponseNorm statistics: scale[b][c] = 1 + gamma[c] * Gx[b][c] / (mean_c Gx[b][:] + 1e-6),
// Gx[b][c] = sqrt(sum over the sample's rows of g[row][c]^2).
int launch_grn_scale(const float* g, const float* gamma, float* scale, float* tmp_gx, int B,
                     int rows_per_sample, int C, hipStream_t stream);

// same statistics from the GEMM epilogue's per-16-row partials (Epilogue::sumsq_out): part [B*groups, C]
int launch_grn_from_partials(const float* part, const float* gamma, float* scale, int B, int groups, int C, hipStream_t stream);

// Token embedding gather + LayerNorm(c_in) + PixelUnshuffle(p): tokens int64 [B,H,W] ->
// out [B*(H/p)*(W/p), c_in*p*p] with channel index c*p*p + dy*p + dx.
int launch_embed_ln_unshuffle(const int64_t* tokens, const float* table, float* out, int B, int H, int W,
                              int c_in, int patch, int num_labels, float eps, hipStream_t stream);

// Sinusoidal timestep embedding + every TimestepBlock mapper in one launch.
// r [B], freqs [c_r/2] (host-computed, torch order), Wcat [total, c_r], bcat [total] -> ts [B, total].
// reps > 1: r holds B distinct samples and ts rows b, b + B, ... (reps of them) receive the same values
int launch_timestep(const float* r, const float* freqs, const float* Wcat, const float* bcat, float* ts,
                    int B, int c_r, int total, float max_positions, float* r_embed_out, int reps, hipStream_t stream);
// x = x*(1+a)+b with [a|b] = ts[b][0:2C] (standalone TimestepBlock).
int launch_scale_shift(float* x, const float* ts, int ts_stride, int64_t rows, int rows_per_sample, int C,
                       hipStream_t stream);

int launch_silu(const float* x, float* y, int64_t n, hipStream_t stream);
int launch_copy_rows(const float* src, int lds, float* dst, int ldd, int64_t rows, int cols, hipStream_t stream);
int launch_axpby(float* x, const float* y, float a, float b, int64_t n, hipStream_t stream);  // x = a*x + b*y
int launch_axpby16(float* x, const float* y, float a, float b, int64_t n, unsigned short* x16, hipStream_t stream);  // + optional bf16 copy of the result

// Attention over [self keys | conditioning keys] (reference src/modules.py:7-19,65-79;
// utils/alter_attention.py:4-43). q/k/v are column blocks of row-major buffers.
struct AttnArgs {
    const float* q; int ldq;          // [B*Lq, ...], head h at columns h*D
    const float* k_self; const float* v_self; int ld_self;   // [B*Lself, ...] or null when Lself == 0
    const float* k_cond; const float* v_cond; int ld_cond;   // [B*Lcond, ...]
    float* out; int ldo;              // [B*Lq, nhead*D]
    int B, nhead, D, Lq, Lself, Lcond;
    float scale;
    const float* key_weights;         // [n_kw] post-softmax multipliers for the LAST n_kw keys, or null
    int n_kw;
    unsigned short* out16;            // optional: the output rounded to bf16 at the same [row, ldo] positions INSTEAD of `out` (opt-in fast mode: feeds the out-projection)
    // opt-in bf16 fast mode, large query counts (attention_bf16_kernel): q / self k / self v as bf16 column blocks of one buffer (leading dimension ld16 elements),
    // both contractions on v_mfma_f32_16x16x16_bf16 (softmax in fp32; the conditioning K / V stay the fp32 cache and are rounded while staged).  q16 == null -> fp32 kernels
    const unsigned short* q16; const unsigned short* k_self16; const unsigned short* v_self16; int ld16;
};
int launch_attention(const AttnArgs& a, hipStream_t stream);

// Sampling tail (reference src/utils.py:45-54): CFG mix, temperature, softmax, categorical draw, renoise.
struct TailArgs {
    const float* logits_c; const float* logits_u;  // [rows, L]; logits_u null -> no CFG
    int64_t rows; int L;
    float cfg, one_minus_cfg, temperature;
    int mode;                     // 0 = categorical, 1 = argmax (T=0 extension)
    const float* noise_q;         // [rows, L] Exp(1) noise (parity mode) or null -> Philox
    uint64_t seed; uint64_t offset;
    int64_t row_offset;           // Philox counters use row + row_offset: a batch shard [lo, hi) passes lo * H * W and draws the noise of its GLOBAL rows
    const uint64_t* seed_ptr;     // optional device-resident seed (added to `seed`): lets a captured HIP graph be replayed with new noise
    const int64_t* row_offset_ptr;  // optional device-resident word added to row_offset: the same graph replayed for another batch shard
    const int64_t* init_noise;    // renoise source or null (no renoise)
    const float* mask_u;          // [rows] U[0,1) (parity mode) or null -> Philox
    float t_next;                 // uniform over the batch inside sample()
    int64_t* tokens_out;          // [rows]
    int64_t* sampled_out;         // [rows] pre-renoise draw (optional, may be null)
};
int launch_sample_tail(const TailArgs& a, hipStream_t stream);
// start tokens of the counter-based mode: out[i] = Philox(seed (+ *seed_ptr), i + row_offset (+ *row_offset_ptr)) % num_labels
int launch_start_tokens(uint64_t seed, const uint64_t* seed_ptr, int64_t row_offset, const int64_t* row_offset_ptr, int num_labels, int64_t n,
                        int64_t* out, hipStream_t stream);

// add_noise (reference src/modules.py:277-283)
int launch_add_noise(const int64_t* x, const float* t, const int64_t* mask_in, const int64_t* random_x,
                     const float* rand_u, uint64_t seed, uint64_t offset, int num_labels, int B,
                     int64_t per_sample, int64_t* x_out, int64_t* mask_out, hipStream_t stream);

// out[i] = ((mask == null || mask[i] != 0) && (flag == null || *flag == 1.0f)) ? a[i] : (b ? b[i] : fill)   (tail.hip)
int launch_select_tokens(const int64_t* a, const int64_t* b, const int64_t* mask, const float* flag, int64_t fill, int64_t n, int64_t* out, hipStream_t stream);

// VQGAN helpers
int launch_codebook_gather(const int64_t* idx, const float* codebook, float* out, int64_t rows, int D, int K,
                           float scale, hipStream_t stream);
// Transposed conv k4 s2 p1 phase gather: for phase (py,px) builds A[rows_in, 4*C] from x [B,H,W,C]
int launch_convT4_gather(const float* x, float* out, int B, int H, int W, int C, int py, int px, hipStream_t stream);
// Conv k4 s2 p1 im2col: x [B,H,W,C] -> out [B*(H/2)*(W/2), 16*C], k index (ky,kx,c)
int launch_conv4s2_im2col(const float* x, float* out, int B, int H, int W, int C, hipStream_t stream);
// image NCHW [B,3,Hp,Wp] -> PixelUnshuffle(2) NHWC [B*(Hp/2)*(Wp/2), 12], channel c*4+dy*2+dx
int launch_img_unshuffle(const float* img, float* out, int B, int C, int Hp, int Wp, hipStream_t stream);
// nearest codebook row (squared L2, first minimum wins)
int launch_vq_nearest(const float* x, const float* codebook, int64_t* idx, float* qe, int64_t rows, int D, int K,
                      hipStream_t stream);
// y[row][c] = x[row][c]*scale[c] + shift[c]  (BatchNorm eval) and NHWC -> NCHW transposes
int launch_affine_cols(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C,
                       hipStream_t stream);
// y = divide ? x / scale : x * scale, with the layout change
int launch_nhwc_to_nchw(const float* x, float* y, int B, int HW, int C, float scale, int divide, hipStream_t stream);
int launch_nchw_to_nhwc(const float* x, float* y, int B, int HW, int C, float scale, int divide, hipStream_t stream);
// generic <=5-d permute-copy used once per tensor when weights are loaded
int launch_permute(const float* src, float* dst, const int64_t* shape, const int* perm, int ndim, hipStream_t stream);
