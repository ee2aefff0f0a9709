// Counter-based noise shared by the sampling tail (tail.hip) and the head GEMM's fused tail epilogue (gemm.hip):
// Philox4x32-10 keyed by (seed, counter), uniform -> Exp(1) / Gumbel helpers.  Both users must draw the SAME numbers for the same
// (seed, global row, label quad, step), so the arithmetic lives in exactly one place.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

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
// (0,1), both ends excluded: 23 random bits + 1/2 is exact in fp32 (24 significant bits), so u lies in [2^-24, 1 - 2^-24] and both logarithms of
// the Gumbel draw stay finite.  (With 24 bits + 1/2 the top value rounds to 1.0f: log(-log 1) = -inf hands that label a +inf score about once
// per 2^24 logits -- found in round 3 by the decision-margin test.)
__device__ __forceinline__ float u01_open(uint32_t bits) { return ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f); }
// [0,1): torch.rand semantics (24-bit mantissa grid)
__device__ __forceinline__ float u01_half_open(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }


// Categorical draw in the log domain (Gumbel-max): token = argmax_i (x_i - log q_i), q ~ Exp(1), i.e. log q = log(-log u).
// == argmax softmax(x) / q == torch.multinomial(softmax(x), 1) in distribution; no exp, no division, no row max needed.
// (hardware log2 for both logarithms: this runs once per LOGIT, 8192 x positions x steps, inside the head GEMM's epilogue)
__device__ __forceinline__ float log_exp1(uint32_t bits) { return __logf(-__logf(u01_open(bits))); }
// The score both tails maximise in the counter-based mode.  ONE definition: the fused (GEMM epilogue) and unfused (tail kernel)
// paths must round identically.  contract(off): no FMA may merge the division's multiply-free result with the subtraction.
// inv_temperature = tail_inv_temperature(T), ONE correctly rounded division per thread instead of one per logit (the counter-based mode promises no bit parity with
// torch's RNG stream, only fused == unfused, which share this function; the torch-noise parity mode keeps the reference's x / T, tail.hip).
__device__ __forceinline__ float tail_inv_temperature(float temperature) { return __fdiv_rn(1.0f, temperature); }
__device__ __forceinline__ float tail_score_gumbel(float logit, float inv_temperature, float log_q) {
    return __fsub_rn(__fmul_rn(logit, inv_temperature), log_q);
}
// first index wins ties (deterministic under any reduction order)
__device__ __forceinline__ void argmax_update(float& best, int& best_i, float score, int idx) {
    if (score > best || (score == best && idx < best_i)) { best = score; best_i = idx; }
}
