// Sampling tail of Paella's sample() loop for gfx950 (reference src/utils.py:45-54,
// src_distributed/utils.py:116-125) and add_noise (reference src/modules.py:277-283).
//
//   l       = l_c*cfg + l_u*(1-cfg)                (two roundings, no FMA: matches torch's two ops)
//   x       = l / T                                (IEEE division, as tensor.div)
//   token   = categorical(softmax(x))              == argmax_i exp(x_i - max)/q_i, q ~ Exp(1)
//             (torch.multinomial(p, 1) is argmax(p / q) with q = empty_like(p).exponential_(1))
//   renoise = u <= t_next ? init_noise : token     (torch.rand_like(x.float()) <= t)
//
// HBM-bound: one 256-thread workgroup per position streams the position's 2 x L logits with 16-byte
// lanes; max / argmax reductions are wave64 shuffles plus one LDS hop across the 4 waves.
// Noise comes either from caller-provided tensors (parity mode: bit-identical draws to torch given the
// same q / u) or from an in-kernel Philox4x32-10 stream keyed by (seed, offset).
#include "common.h"
#include "test_hooks.h"
#include "philox.h"
#include <math.h>

#pragma clang fp contract(off)

__device__ __forceinline__ float mix_logit(float lc, float lu, float cfg, float omc, bool has_u) {
    return has_u ? __fadd_rn(__fmul_rn(lc, cfg), __fmul_rn(lu, omc)) : lc;
}

// renoise (src/utils.py:54 -> src/modules.py:277-283 with random_x = init_noise): u <= t_next ? init_noise : token
__device__ __forceinline__ int64_t tail_row_offset(const TailArgs& a) { return a.row_offset + (a.row_offset_ptr ? *a.row_offset_ptr : 0); }

__device__ __forceinline__ int64_t renoise_token(const TailArgs& a, uint64_t seed, int64_t row, int64_t tok) {
    if (a.init_noise) {
        float u;
        if (a.mask_u) {
            u = a.mask_u[row];
        } else {
            uint32_t rb[4];
            philox4x32(seed ^ 0x5bd1e9955bd1e995ull, (uint64_t)(row + tail_row_offset(a)), a.offset, rb);
            u = u01_half_open(rb[0]);
        }
        if (u <= a.t_next) tok = a.init_noise[row];
    }
    return tok;
}

__global__ __launch_bounds__(256) void sample_tail_kernel(TailArgs a) {
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L, L4 = L >> 2;
    const float* lc = a.logits_c + row * L;
    const float* lu = a.logits_u ? a.logits_u + row * L : nullptr;
    const bool has_u = lu != nullptr;
    const bool argmax_mode = a.mode == 1;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
    const float* nq = a.noise_q ? a.noise_q + row * L : nullptr;
    const int64_t row_off = tail_row_offset(a);

    // pass 1 (explicit-noise parity mode only): max of x = mix / T for the softmax numerator exp(x - max).
    // The argmax and the counter-based mode never need it: argmax(x - log q) is invariant to a per-row shift.
    float mx = 0.f;
    if (nq && !argmax_mode) {
        mx = -INFINITY;
        for (int i4 = tid; i4 < L4; i4 += 256) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(lc + i4 * 4);
            const f32x4 u = has_u ? *reinterpret_cast<const f32x4*>(lu + i4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, __fdiv_rn(mix_logit(c[e], u[e], a.cfg, a.one_minus_cfg, has_u), a.temperature));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) red_v[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red_v[0], red_v[1]), fmaxf(red_v[2], red_v[3]));
        __syncthreads();
    }

    // pass 2: best score (first index wins ties)
    const float inv_t = tail_inv_temperature(a.temperature);  // counter-based mode (philox.h: tail_score_gumbel)
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int i4 = tid; i4 < L4; i4 += 256) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(lc + i4 * 4);
        const f32x4 u = has_u ? *reinterpret_cast<const f32x4*>(lu + i4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 q = f32x4{1.f, 1.f, 1.f, 1.f};
        if (!argmax_mode) {
            if (nq) {
                q = *reinterpret_cast<const f32x4*>(nq + i4 * 4);
            } else {
                uint32_t rb[4];
                philox4x32(seed, (uint64_t)(row + row_off) * L4 + i4, a.offset, rb);
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = log_exp1(rb[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = mix_logit(c[e], u[e], a.cfg, a.one_minus_cfg, has_u);
            float score;
            if (argmax_mode) {
                score = x;
            } else if (nq) {  // parity mode: the reference's arithmetic, softmax numerator over Exp(1) noise
                x = __fdiv_rn(x, a.temperature);
                score = __fdiv_rn(expf(__fsub_rn(x, mx)), q[e]);
            } else {          // counter-based noise: the same draw in the log domain (Gumbel-max); identical arithmetic in the
                score = tail_score_gumbel(x, inv_t, q[e]);  // head GEMM's fused tail epilogue (gemm.hip)
            }
            argmax_update(best, best_i, score, i4 * 4 + e);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        argmax_update(best, best_i, ov, oi);
    }
    if (lane == 0) { red_v[wave] = best; red_i[wave] = best_i; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) argmax_update(best, best_i, red_v[w], red_i[w]);
        if (best_i == 0x7fffffff) best_i = 0;  // all-NaN row
        int64_t tok = best_i;
        if (a.sampled_out) a.sampled_out[row] = tok;
        a.tokens_out[row] = renoise_token(a, seed, row, tok);
    }
}

// Second half of the FUSED tail: the head GEMM's epilogue (gemm.hip, TAIL instantiations) left, per row and column tile, the best
// (score, label) of that tile; pick the row's winner (first index wins ties -> identical to the one-kernel tail under any
// reduction order), renoise, store the token.  One wave per row: lane t reads tile t (coalesced), xor-shuffle argmax.
__global__ __launch_bounds__(256) void tail_finalize_kernel(TailArgs a, const float* __restrict__ part_score, const int* __restrict__ part_idx,
                                                            int tiles_n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    const float* ps = part_score + row * tiles_n;
    const int* pi = part_idx + row * tiles_n;
    for (int t = lane; t < tiles_n; t += 64) argmax_update(best, best_i, ps[t], pi[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        argmax_update(best, best_i, ov, oi);
    }
    if (lane == 0) {
        const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
        if (best_i == 0x7fffffff) best_i = 0;
        int64_t tok = best_i;
        if (a.sampled_out) a.sampled_out[row] = tok;
        a.tokens_out[row] = renoise_token(a, seed, row, tok);
    }
}

int launch_tail_finalize(const TailArgs& a, const float* part_score, const int* part_idx, int tiles_n, hipStream_t st) {
    if (a.rows <= 0) return PAELLA_OK;
    hipLaunchKernelGGL(tail_finalize_kernel, dim3((unsigned)((a.rows + 3) / 4)), dim3(256), 0, st, a, part_score, part_idx, tiles_n);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

int launch_sample_tail(const TailArgs& a, hipStream_t st) {
    if (a.rows <= 0) return PAELLA_OK;
    if (a.L & 3) { paella_set_error("sample_tail: num_labels %% 4 != 0"); return PAELLA_ERR_ARG; }
    if (a.rows > 0x7fffffff) { paella_set_error("sample_tail: too many rows"); return PAELLA_ERR_ARG; }
    hipLaunchKernelGGL(sample_tail_kernel, dim3((unsigned)a.rows), dim3(256), 0, st, a);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// Start tokens of the counter-based noise mode: the reference draws randint(0, num_labels, (B, H, W)) from torch's global generator
// (src/utils.py:37), a stream that cannot be sharded.  Here token i of the GLOBAL grid is a function of (seed, i) alone, so a batch
// shard draws exactly the start tokens the unsharded call gives its rows, at O(shard) cost, and -- seed and row offset being optional
// device-resident words -- from inside a captured graph.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void start_tokens_kernel(uint64_t seed, const uint64_t* __restrict__ seed_ptr, int64_t row_offset,
                                                           const int64_t* __restrict__ row_offset_ptr, int num_labels, int64_t n,
                                                           int64_t* __restrict__ out) {
    const uint64_t s = (seed + (seed_ptr ? *seed_ptr : 0ull)) ^ 0x9e3779b97f4a7c15ull;
    const int64_t off = row_offset + (row_offset_ptr ? *row_offset_ptr : 0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t rb[4];
        philox4x32(s, (uint64_t)(i + off), ~0ull, rb);
        out[i] = (int64_t)((((uint64_t)rb[0] << 32) | rb[1]) % (uint64_t)num_labels);
    }
}
int launch_start_tokens(uint64_t seed, const uint64_t* seed_ptr, int64_t row_offset, const int64_t* row_offset_ptr, int num_labels, int64_t n,
                        int64_t* out, hipStream_t st) {
    if (n <= 0) return PAELLA_OK;
    if (!out || num_labels <= 0 || row_offset < 0) { paella_set_error("start_tokens: bad arguments"); return PAELLA_ERR_ARG; }
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(start_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, seed_ptr, row_offset, row_offset_ptr, num_labels, n, out);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// add_noise: mask = (U[0,1) <= t[b]).long(); x*(1-mask) + random_x*mask   (int64 arithmetic as written)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_noise_kernel(const int64_t* __restrict__ x, const float* __restrict__ t,
                                                        const int64_t* __restrict__ mask_in, const int64_t* __restrict__ random_x,
                                                        const float* __restrict__ rand_u, uint64_t seed, uint64_t offset,
                                                        int num_labels, int64_t total, int64_t per_sample,
                                                        int64_t* __restrict__ x_out, int64_t* __restrict__ mask_out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        uint32_t rb[4] = {0, 0, 0, 0};
        if ((!mask_in && !rand_u) || !random_x) philox4x32(seed, (uint64_t)i, offset, rb);
        int64_t m;
        if (mask_in) m = mask_in[i];
        else {
            const float u = rand_u ? rand_u[i] : u01_half_open(rb[0]);
            m = (u <= t[i / per_sample]) ? 1 : 0;
        }
        const int64_t rx = random_x ? random_x[i] : (int64_t)((((uint64_t)rb[1] << 32) | rb[2]) % (uint64_t)num_labels);
        x_out[i] = x[i] * (1 - m) + rx * m;
        if (mask_out) mask_out[i] = m;
    }
}

int launch_add_noise(const int64_t* x, const float* t, const int64_t* mask_in, const int64_t* random_x,
                     const float* rand_u, uint64_t seed, uint64_t offset, int num_labels, int B, int64_t per_sample,
                     int64_t* x_out, int64_t* mask_out, hipStream_t st) {
    const int64_t total = (int64_t)B * per_sample;
    if (total <= 0) return PAELLA_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, t, mask_in, random_x, rand_u, seed,
                       offset, num_labels, total, per_sample, x_out, mask_out);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// Token select: out[i] = keep(i) ? a[i] : (b ? b[i] : fill), keep(i) = (mask == null || mask[i] != 0) && (flag == null || *flag == 1.0f).
// Two users, both integer elementwise work on the [B, H, W] grid that used to go through ATen: the inpainting wrapper re-imposes the known tokens
// (`out * mask + tokens * (1 - mask)`, paella_amd/editing.py -- an extension of the recipe src/modules.py:277-283 + src_distributed/utils.py:97-109) and the
// batch-sharded sampler replaces a receiver's tokens by -1 when the conditioning broadcast carried a zero validity flag (paella_amd/dist.py).  `flag` is a
// DEVICE word, so neither needs a host round trip and both can be captured in a HIP graph.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_tokens_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b, const int64_t* __restrict__ mask,
                                                            const float* __restrict__ flag, int64_t fill, int64_t n, int64_t* __restrict__ out) {
    const bool ok = flag ? (*flag == 1.0f) : true;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const bool keep = ok && (mask ? mask[i] != 0 : true);
        out[i] = keep ? a[i] : (b ? b[i] : fill);
    }
}
int launch_select_tokens(const int64_t* a, const int64_t* b, const int64_t* mask, const float* flag, int64_t fill, int64_t n, int64_t* out, hipStream_t st) {
    if (n <= 0) return PAELLA_OK;
    if (!a || !out) { paella_set_error("select_tokens: null argument"); return PAELLA_ERR_ARG; }
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(select_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, mask, flag, fill, n, out);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// test hook (test_hooks.h): the scores the counter-based tail maximises, written out -- score[row][i] = mix(l_c, l_u)[i] / T - log q_i
// with exactly the kernels' arithmetic and Philox counters.  Lets a test classify a differing token by the decision margin
// (top-1 minus top-2 score) instead of bounding a mismatch count.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tail_scores_kernel(TailArgs a, float* __restrict__ scores) {
    const int64_t row = blockIdx.x;
    const int L = a.L, L4 = L >> 2;
    const float* lc = a.logits_c + row * L;
    const float* lu = a.logits_u ? a.logits_u + row * L : nullptr;
    const bool has_u = lu != nullptr;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
    const int64_t row_off = tail_row_offset(a);
    for (int i4 = threadIdx.x; i4 < L4; i4 += 256) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(lc + i4 * 4);
        const f32x4 u = has_u ? *reinterpret_cast<const f32x4*>(lu + i4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        uint32_t rb[4];
        philox4x32(seed, (uint64_t)(row + row_off) * L4 + i4, a.offset, rb);
        f32x4 s;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = tail_score_gumbel(mix_logit(c[e], u[e], a.cfg, a.one_minus_cfg, has_u), tail_inv_temperature(a.temperature), log_exp1(rb[e]));
        *reinterpret_cast<f32x4*>(scores + row * L + i4 * 4) = s;
    }
}
extern "C" int paella_test_tail_scores(const float* logits_c, const float* logits_u, int64_t rows, int L, float cfg, float one_minus_cfg,
                                       float temperature, uint64_t seed, uint64_t offset, int64_t row_offset, float* scores_out, void* stream) {
    if (!logits_c || !scores_out || (L & 3) || rows <= 0 || rows > 0x7fffffff || !(temperature > 0.f)) { paella_set_error("tail_scores: bad arguments"); return PAELLA_ERR_ARG; }
    TailArgs a = {};
    a.logits_c = logits_c; a.logits_u = logits_u; a.rows = rows; a.L = L; a.cfg = cfg; a.one_minus_cfg = one_minus_cfg; a.temperature = temperature;
    a.seed = seed; a.offset = offset; a.row_offset = row_offset;
    hipLaunchKernelGGL(tail_scores_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, scores_out);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// measurement hook: a chain of n dependent, nearly empty kernels on `stream` (launch-boundary floor of this box)
// ---------------------------------------------------------------------------
__global__ void chain_kernel(float* p, int bytes4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bytes4) p[i] += 1.0f;
}
extern "C" int paella_test_launch_chain(float* buf, int n_elems, int blocks, int n_launches, void* stream) {
    for (int i = 0; i < n_launches; ++i) hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, n_elems);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
