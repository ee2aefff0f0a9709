// Device-side epilogue shared by the GEMM kernels (gemm.hip, gemm_bf16.hip).
#pragma once
#include "common.h"

// Sum over the 16 lanes of a DPP row (the 16 rows r16 of a fragment block), every lane receiving the total: the xor butterfly 1, 2, 4, 8 -- bit for bit, because after
// the quad steps all lanes of a quad hold the same value, so the mirror partners (7 - i, 15 - i) carry exactly what lanes i ^ 4, i ^ 8 do -- as four v_add_f32_dpp
// instead of four ds_bpermute_b32 round trips through the LDS pipe (__shfl_xor compiles to ds_bpermute on gfx950; the GRN statistics of an MLP's first GEMM need 16
// of these reductions per 16x16 accumulator block, 256 per lane and 256x128 tile).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1, 0, 3, 2]  == xor 1
    v += dpp_mov<0x4E>(v);   // quad_perm [2, 3, 0, 1]  == xor 2
    v += dpp_mov<0x141>(v);  // row_half_mirror         == xor 4 (quads already uniform)
    v += dpp_mov<0x140>(v);  // row_mirror              == xor 8 (halves already uniform)
    return v;
}

// GELU(erf) of the exact path: 0.5 x (1 + erf(x / sqrt 2)) with erf(a) = sign . (1 - 2^-P(|a|)), P = a degree-8 fit of -log2 erfc on [0, 3.92] (erfc(3.92) < 2^-25: the
// clamp IS erf = 1 in fp32), ONE range, no branch: 15 instructions + one v_exp_f32.  libm's erff is two ranges behind a divergent branch (~37 instructions, both sides
// executed by a wave whose lanes straddle |x| = 1 -- every wave of an MLP epilogue).  GELU needs erf to ABSOLUTE accuracy only (it enters as 1 + erf), which is what the
// one-range form gives: max |gelu - exact| = 5.5e-7 over [-8, 8] against 6.8e-7 for 0.5 x (1 + erff(.)) evaluated in fp32 (both are the final roundings at large |x|;
// fit and comparison: tools/fit_gelu.py).
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fminf(fabsf(x) * 0.70710678118654752440f, 3.92f);
    float q = 3.501229730e-05f;
    q = fmaf(q, a, -3.597551840e-04f);
    q = fmaf(q, a, 1.208558562e-03f);
    q = fmaf(q, a, 1.240070444e-03f);
    q = fmaf(q, a, -2.866797522e-02f);
    q = fmaf(q, a, 1.486749798e-01f);
    q = fmaf(q, a, 9.183741808e-01f);
    q = fmaf(q, a, 1.627911687e+00f);
    const float e = copysignf(1.0f - __builtin_amdgcn_exp2f(-(q * a)), x);
    const float h = 0.5f * x;
    return fmaf(h, e, h);
}

// GELU of the opt-in bf16 fast mode's epilogues (bf16 operands: the accumulator already carries ~1e-3 relative operand-rounding error and the MLP hidden tensor it
// feeds is stored as bf16, 2^-9 relative).  libm's erff is ~37 VALU instructions with a divergent branch -- at bf16 MFMA rates that is AS LONG as the K = 1280 main loop
// of an MLP's first GEMM and 3x the K = 384 one.  Branch-free, 12 instructions, no transcendental:
//     gelu(x) = max(x, 0) - |x| Q(|x|),   Q(a) = 1 - Phi(a) ~ (4 - a)+ . R(a),  R = degree-6 minimax fit on [0, 4] (weighted by the error it causes in gelu)
// max |gelu_fast - gelu| = 1.27e-4 over all x (equi-oscillating; |x| >= 4 returns max(x, 0), true value differs by <= 4 Q(4) = 1.27e-4).  The exact fp32 path never uses it.
__device__ __forceinline__ float gelu_fast(float x) {
    const float a = fabsf(x);
    const float d = fmaxf(4.0f - a, 0.0f);
    float r = -4.582215843e-05f;
    r = fmaf(r, a, 1.208787551e-03f);
    r = fmaf(r, a, -1.008340903e-02f);
    r = fmaf(r, a, 3.467543423e-02f);
    r = fmaf(r, a, -3.520498052e-02f);
    r = fmaf(r, a, -6.194137782e-02f);
    r = fmaf(r, a, 1.242401227e-01f);
    return fmaf(-(a * d), r, fmaxf(x, 0.0f));
}

// bias -> activation -> alpha -> residual -> timestep scale/shift (FASTG: the bf16-operand kernels' GELU)
template <bool FASTG = false>
__device__ __forceinline__ f32x4 epilogue_apply(const Epilogue& ep, int N, int m, int n, f32x4 v) {
    if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + n);
    if (ep.act == ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = FASTG ? gelu_fast(v[i]) : gelu_erf(v[i]);
    }
    if (ep.alpha != 1.0f) v *= ep.alpha;
    if (ep.residual) v += *reinterpret_cast<const f32x4*>(ep.residual + (size_t)m * ep.ldr + n);
    if (ep.ts) {
        const float* t = ep.ts + (size_t)fast_div((unsigned)m, ep.rps_div) * ep.ts_stride;
        f32x4 a = *reinterpret_cast<const f32x4*>(t + n);
        f32x4 b = *reinterpret_cast<const f32x4*>(t + N + n);
        v = v * (1.0f + a) + b;
    }
    return v;
}

__device__ __forceinline__ void epilogue_write(const Epilogue& ep, float* __restrict__ C, int ldc, int m, int n, f32x4 v) {
    if (ep.store_mode == STORE_PLAIN) {
        size_t orow = m;
        if (ep.remap_in > 0) orow = (size_t)(m / ep.remap_in) * ep.remap_out + (m % ep.remap_in) + ep.remap_off;
        if (C) *reinterpret_cast<f32x4*>(C + orow * ldc + n) = v;
        if (ep.c16) {  // kernel-uniform: bf16 copy (RNE, v_cvt_pk_bf16_f32) for a consuming bf16 GEMM -- opt-in fast mode only
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            *reinterpret_cast<bf16x4_t*>(ep.c16 + orow * ldc + n) = __builtin_convertvector(v, bf16x4_t);
        }
    } else {
        const int hw = ep.sH * ep.sW;
        const int b = m / hw;
        const int rem = m - b * hw;
        const int y = rem / ep.sW;
        const int x = rem - y * ep.sW;
        if (ep.store_mode == STORE_D2S) {
            const int seg = n / ep.sC;
            const int co = n - seg * ep.sC;
            const int dy = seg / ep.n_seg_x;
            const int dx = seg - dy * ep.n_seg_x;
            const size_t orow = ((size_t)b * (2 * ep.sH) + 2 * y + dy + ep.py) * (2 * ep.sW) + 2 * x + dx + ep.px;
            *reinterpret_cast<f32x4*>(C + orow * ldc + co) = v;
        } else {  // STORE_PIXSHUF_NCHW: n = c*4 + dy*2 + dx -> out[b][c][2y+dy][2x+dx]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nn = n + i;
                const int c = nn >> 2, dy = (nn >> 1) & 1, dx = nn & 1;
                C[(((size_t)b * ep.sC + c) * (2 * ep.sH) + 2 * y + dy) * (2 * ep.sW) + 2 * x + dx] = v[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// LayerNorm-on-load row statistics (producer: Epilogue::rowstat_out, consumer: GemmArgs::ln_stats), [M, N/16, 2].
// Per row and 16-column block the producer leaves (sum, M2) with M2 = sum (v - block mean)^2 -- CENTRED partials, not (sum, sum of squares):
// the consumer combines them with the parallel-variance formula  M2_total = sum_j M2_j + sum_j s_j^2 / 16 - S^2 / K  in fp64, whose cancellation
// acts on the EXACTLY representable block sums only.  Relative error of the variance ~ eps * |mean| / std (linear), where the one-pass
// E[x^2] - mean^2 over fp32 partial sums of squares loses eps * (mean / std)^2 (round 3).
// ---------------------------------------------------------------------------
// v = this lane's 4 of the block's 16 values; the block's other 12 live in lanes +-16, +-32 (the kq = 0..3 lanes of a fragment row)
__device__ __forceinline__ void rowstat_block(const f32x4 v, float& s, float& m2) {
    s = (v[0] + v[1]) + (v[2] + v[3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mb = s * 0.0625f;
    const float d0 = v[0] - mb, d1 = v[1] - mb, d2 = v[2] - mb, d3 = v[3] - mb;
    m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    m2 += __shfl_xor(m2, 16, 64);
    m2 += __shfl_xor(m2, 32, 64);
}
struct RowStatAcc {
    double S = 0.0, Q = 0.0, M = 0.0;  // sum of block sums, sum of block sums squared, sum of block M2
    __device__ __forceinline__ void add(float s, float m2) { S += (double)s; Q += (double)s * (double)s; M += (double)m2; }
    __device__ __forceinline__ void finish(int K, float eps, float& mu, float& rstd) const {
        const double mean = S / (double)K;
        const double var = (M + Q * 0.0625 - S * mean) / (double)K;
        mu = (float)mean;
        rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
};
// LayerNorm folded into the consuming GEMM's epilogue, rstd * (acc - mu * wsum[n]), cancels when |mu| >> std: its error is ~ eps * sqrt(K) * |mu| / std
// relative to the normalised product (measured on K = 1280: 6e-6 * ratio, i.e. 7e-5 at 10, 6e-4 at 100, 5.6e-3 at 1000 against a flat 5e-6 for the
// operand-side form: profiles/r04_ln_fold_error_curve.txt).  Above this ratio a 16-row fragment block switches to normalising
// its operand fragments instead ((a - mu) * rstd before the MFMAs; error independent of the ratio).  A wave-uniform decision per 16-row block that
// depends only on the rows' statistics, so every workgroup that shares a tile takes the same path.
static constexpr float kLnFoldMaxRatio = 4.0f;  // default of GemmArgs::ln_fold_ratio

__device__ __forceinline__ void epilogue_store(const Epilogue& ep, float* __restrict__ C, int ldc, int N,
                                               int m, int n, f32x4 v) {
    epilogue_write(ep, C, ldc, m, n, epilogue_apply(ep, N, m, n, v));
}
