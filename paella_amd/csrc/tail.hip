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
#include <math.h>

#pragma clang fp contract(off)

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// (0,1): 24 random bits, never 0 -> -log(u) finite and > 0
__device__ __forceinline__ float u01_open(uint32_t bits) { return ((float)(bits >> 8) + 0.5f) * (1.0f / 16777216.0f); }
// [0,1): torch.rand semantics (24-bit mantissa grid)
__device__ __forceinline__ float u01_half_open(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float mix_logit(float lc, float lu, float cfg, float omc, bool has_u) {
    return has_u ? __fadd_rn(__fmul_rn(lc, cfg), __fmul_rn(lu, omc)) : lc;
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

    // pass 1: max of x = mix / T
    float mx = -INFINITY;
    for (int i4 = tid; i4 < L4; i4 += 256) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(lc + i4 * 4);
        const f32x4 u = has_u ? *reinterpret_cast<const f32x4*>(lu + i4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = mix_logit(c[e], u[e], a.cfg, a.one_minus_cfg, has_u);
            if (!argmax_mode) x = __fdiv_rn(x, a.temperature);
            mx = fmaxf(mx, x);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red_v[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_v[0], red_v[1]), fmaxf(red_v[2], red_v[3]));
    __syncthreads();

    // pass 2: best score (first index wins ties)
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    const float* nq = a.noise_q ? a.noise_q + row * L : nullptr;
    for (int i4 = tid; i4 < L4; i4 += 256) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(lc + i4 * 4);
        const f32x4 u = has_u ? *reinterpret_cast<const f32x4*>(lu + i4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 q = f32x4{1.f, 1.f, 1.f, 1.f};
        if (!argmax_mode) {
            if (nq) {
                q = *reinterpret_cast<const f32x4*>(nq + i4 * 4);
            } else {
                uint32_t rb[4];
                philox4x32(seed, (uint64_t)(row + a.row_offset) * L4 + i4, a.offset, rb);
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = __logf(-logf(u01_open(rb[e])));  // log of an Exp(1) variate: argmax(p/q) == argmax(log p - log q)
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
            } else {          // counter-based noise: the same draw in the log domain (Gumbel-max), no exp and no second division
                x = __fdiv_rn(x, a.temperature);
                score = (x - mx) - q[e];
            }
            const int idx = i4 * 4 + e;
            if (score > best || (score == best && idx < best_i)) { best = score; best_i = idx; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    if (lane == 0) { red_v[wave] = best; red_i[wave] = best_i; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (red_v[w] > best || (red_v[w] == best && red_i[w] < best_i)) { best = red_v[w]; best_i = red_i[w]; }
        if (best_i == 0x7fffffff) best_i = 0;  // all-NaN row
        int64_t tok = best_i;
        if (a.sampled_out) a.sampled_out[row] = tok;
        if (a.init_noise) {
            float u;
            if (a.mask_u) {
                u = a.mask_u[row];
            } else {
                uint32_t rb[4];
                philox4x32(seed ^ 0x5bd1e9955bd1e995ull, (uint64_t)(row + a.row_offset), a.offset, rb);
                u = u01_half_open(rb[0]);
            }
            if (u <= a.t_next) tok = a.init_noise[row];
        }
        a.tokens_out[row] = tok;
    }
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
