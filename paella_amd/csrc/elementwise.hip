// HBM-bound kernels of the Paella hot path for gfx950: LayerNorm, VQGAN depthwise 3x3 conv, GRN statistics,
// token-embedding gather, timestep embedding, layout shuffles.  All activations are NHWC fp32
// ([rows = B*h*w, C] row-major), so every wave streams whole channel rows with 16-byte lanes and all
// per-position reductions are wave64 shuffles (no LDS, no barriers) -- one wave per position.
#include "common.h"
#include "gemm_device.h"

#define WAVES_PER_BLOCK 4

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// bf16 outputs of the opt-in fast mode (the A operands of bf16 GEMMs): round-to-nearest-even, 4 values = 8 bytes
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void st4h(unsigned short* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }

// Normalise NV float4 per lane held in registers: two-pass mean / biased variance (torch LayerNorm semantics).
template <int NV>
__device__ __forceinline__ void ln_regs(f32x4 (&v)[NV], int C4, int lane, int C, float eps, float mul, float add) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < C4) s += hsum4(v[i]);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < C4) {
            f32x4 d = v[i] - mean;
            d = d * d;
            q += hsum4(d);
        }
    const float var = wave_sum(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (v[i] - mean) * rstd * mul + add;
}

// ---------------------------------------------------------------------------
// LayerNorm (reference src/modules.py:22-27 LayerNorm2d; src/vqgan.py:35-39 norm + gamma affine)
// ---------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned short* __restrict__ y16,
                                                                        int64_t rows, int C, float eps, float g_mul,
                                                                        float g_add, int s2d, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C4 = C >> 2;
    f32x4 v[NV];
    const float* xr = x + row * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        v[i] = c4 < C4 ? ld4(xr + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    ln_regs<NV>(v, C4, lane, C, eps, g_mul, g_add);
    int64_t yo;  // element offset of this row's output
    if (s2d) {
        const int64_t hw = (int64_t)H * W;
        const int64_t b = row / hw;
        const int rem = (int)(row - b * hw);
        const int yy = rem / W, xx = rem - yy * W;
        const int64_t orow = (b * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1);
        yo = orow * (4 * (int64_t)C) + ((yy & 1) * 2 + (xx & 1)) * C;
    } else {
        yo = row * C;
    }
    if (y16) {  // kernel-uniform: the bf16 copy for a consuming bf16 GEMM (y may be null then)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < C4) st4h(y16 + yo + c4 * 4, v[i]);
        }
    }
    if (y) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < C4) st4(y + yo + c4 * 4, v[i]);
        }
    }
}

#define DISPATCH_NV(C4, ...)                                                       \
    do {                                                                           \
        const int _nv = ((C4) + 63) / 64;                                          \
        if (_nv <= 1) { constexpr int NV = 1; __VA_ARGS__; }                       \
        else if (_nv <= 2) { constexpr int NV = 2; __VA_ARGS__; }                  \
        else if (_nv <= 3) { constexpr int NV = 3; __VA_ARGS__; }                  \
        else if (_nv <= 5) { constexpr int NV = 5; __VA_ARGS__; }                  \
        else if (_nv <= 8) { constexpr int NV = 8; __VA_ARGS__; }                  \
        else if (_nv <= 16) { constexpr int NV = 16; __VA_ARGS__; }                \
        else { paella_set_error("channel count %d too large", (C4) * 4); return PAELLA_ERR_ARG; } \
    } while (0)

int launch_layernorm(const float* x, float* y, int64_t rows, int C, float eps, float g_mul, float g_add, int s2d,
                     int H, int W, hipStream_t st) {
    return launch_layernorm16(x, y, nullptr, rows, C, eps, g_mul, g_add, s2d, H, W, st);
}
int launch_layernorm16(const float* x, float* y, unsigned short* y16, int64_t rows, int C, float eps, float g_mul, float g_add, int s2d,
                       int H, int W, hipStream_t st) {
    if (rows <= 0) return PAELLA_OK;
    if (C & 3) { paella_set_error("layernorm: C %% 4 != 0 (C=%d)", C); return PAELLA_ERR_ARG; }
    if (s2d && ((H & 1) || (W & 1))) { paella_set_error("layernorm s2d: odd grid %dx%d", H, W); return PAELLA_ERR_ARG; }
    const unsigned blocks = (unsigned)((rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    DISPATCH_NV(C >> 2, hipLaunchKernelGGL((layernorm_kernel<NV>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, st, x, y, y16,
                                           rows, C, eps, g_mul, g_add, s2d, H, W));
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// Row statistics of a LayerNorm-consuming GEMM, finished ONCE per row instead of once per workgroup: from the producing epilogue's per-16-column
// (sum, centred M2) partials (gemm_device.h) to out[row] = (mean, rstd, mean - (float)mean, |mean| * rstd).  Used for the throughput regime
// (thousands of rows: every one of the N / 64 workgroups of a tile row would otherwise re-derive the same 64 rows in its ramp -- LayerNorm-consuming
// GEMMs ran 8-10 % behind plain ones); the batch-1 launches keep the in-kernel derivation (an extra launch costs more than it saves there).
// 16 lanes per row, fp64 combination, fixed order.
// bf16 fast mode (A16 != null): a row whose |mean| / std exceeds `ratio` loses its digits in the bf16 COPY the consuming GEMM multiplies (the copy's rounding is
// ratio * 2^-9 of a standard deviation per element and the folded LayerNorm cannot undo it) -- such a row of A16 is rewritten here as bf16((x - mean) * rstd) from
// the fp32 row, and the consumer skips the fold for it (gemm_nt_kernel: ln_pre).  Ordinary activations never trip it (|mean| / std <= 0.07 in the model).
__global__ __launch_bounds__(256) void ln_rowstat_finalize_kernel(const float* __restrict__ stats, int nblk, int K, float eps, f32x4* __restrict__ out, int64_t M,
                                                                  const float* __restrict__ A32, unsigned short* __restrict__ A16, int lda, float ratio,
                                                                  unsigned* __restrict__ guard_count) {
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    if (row >= M) return;
    const float* st = stats + (size_t)row * nblk * 2;
    RowStatAcc acc;
    for (int j = l16; j < nblk; j += 16) acc.add(st[2 * j], st[2 * j + 1]);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { acc.S += __shfl_xor(acc.S, o, 64); acc.Q += __shfl_xor(acc.Q, o, 64); acc.M += __shfl_xor(acc.M, o, 64); }
    float mu, rs;
    acc.finish(K, eps, mu, rs);  // (all 16 lanes hold the same totals after the butterfly)
    const float mu_lo = (float)(acc.S / (double)K - (double)mu);
    if (l16 == 0) out[row] = f32x4{mu, rs, mu_lo, fabsf(mu) * rs};
    if (A16 && fabsf(mu) * rs > ratio) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        const float* src = A32 + (size_t)row * lda;
        unsigned short* dst = A16 + (size_t)row * lda;
        for (int k = l16 * 4; k < K; k += 64) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((v[e] - mu) - mu_lo) * rs;
            *reinterpret_cast<bf16x4_t*>(dst + k) = __builtin_convertvector(v, bf16x4_t);
        }
        if (guard_count && l16 == 0) atomicAdd(guard_count, 1u);  // test hook only (null in the product)
    }
}

int launch_ln_rowstat_finalize(const float* stats, int nblk, int K, float eps, float* out4, int64_t M, const float* A32, unsigned short* A16, int lda, float ratio,
                               unsigned* guard_count, hipStream_t st) {
    if (M <= 0) return PAELLA_OK;
    if (A16 && (!A32 || (K & 3) || (lda & 3))) { paella_set_error("ln_rowstat_finalize: the bf16 rewrite needs the fp32 rows and K, lda %% 4 == 0"); return PAELLA_ERR_ARG; }
    hipLaunchKernelGGL(ln_rowstat_finalize_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, stats, nblk, K, eps, reinterpret_cast<f32x4*>(out4), M, A32, A16, lda, ratio,
                       guard_count);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// (the UNet ResBlock front half -- depthwise 3x3 + LayerNorm -- lives in dwconv.hip)

// ---------------------------------------------------------------------------
// VQGAN ResBlock depthwise half (reference src/vqgan.py:11-14,38):
//   y = x + (Conv2d(k=3, groups=C)(ReplicationPad2d(1)(xt)) + bias) * gamma2
// ---------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void dwconv_res_kernel(const float* __restrict__ x,
                                                                         const float* __restrict__ xt,
                                                                         const float* __restrict__ w,
                                                                         const float* __restrict__ bias,
                                                                         float* __restrict__ y, int B, int H, int W, int C,
                                                                         float gamma2) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t total = (int64_t)B * H * W;
    if (pos >= total) return;
    const int C4 = C >> 2;
    const int xx = (int)(pos % W);
    const int yy = (int)((pos / W) % H);
    const int64_t base = pos - (int64_t)yy * W - xx;  // (b, 0, 0)
    f32x4 acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        acc[i] = c4 < C4 ? ld4(bias + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int sy = min(max(yy + ky - 1, 0), H - 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sx = min(max(xx + kx - 1, 0), W - 1);
            const int64_t npos = base + (int64_t)sy * W + sx;
            const int tap = ky * 3 + kx;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c4 = lane + i * 64;
                if (c4 < C4) acc[i] += ld4(xt + npos * C + c4 * 4) * ld4(w + tap * C + c4 * 4);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < C4) st4(y + pos * C + c4 * 4, ld4(x + pos * C + c4 * 4) + acc[i] * gamma2);
    }
}

int launch_dwconv_res(const float* x, const float* xt, const float* w, const float* bias, float* y, int B, int H,
                      int W, int C, float gamma2, hipStream_t st) {
    const int64_t total = (int64_t)B * H * W;
    if (total <= 0) return PAELLA_OK;
    if (C & 3) { paella_set_error("dwconv_res: C %% 4 != 0"); return PAELLA_ERR_ARG; }
    const unsigned blocks = (unsigned)((total + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    DISPATCH_NV(C >> 2, hipLaunchKernelGGL((dwconv_res_kernel<NV>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, st, x, xt, w,
                                           bias, y, B, H, W, C, gamma2));
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// GlobalResponseNorm statistics (reference src/modules.py:37-40).  The apply step
// gamma*(x*Nx)+beta+x == x*(1+gamma*Nx)+beta is folded into the next GEMM's A-operand load.
// Deterministic: fixed-order partial sums, no atomics.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grn_sumsq_kernel(const float* __restrict__ g, float* __restrict__ gx, int rows_per_sample,
                                                        int C) {
    // block = 64 channels x 4 row groups; grid = (ceil(C/64), B)
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int b = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        const float* p = g + ((size_t)b * rows_per_sample) * C + c;
        for (int r = rg; r < rows_per_sample; r += 4) {
            const float v = p[(size_t)r * C];
            s += v * v;
        }
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C) gx[(size_t)b * C + c] = sqrtf((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

__global__ __launch_bounds__(256) void grn_finalize_kernel(const float* __restrict__ gx, const float* __restrict__ gamma,
                                                           float* __restrict__ scale, int C) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    const float* p = gx + (size_t)b * C;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += p[c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float denom = red[0] / (float)C + 1e-6f;
    for (int c = threadIdx.x; c < C; c += 256) scale[(size_t)b * C + c] = 1.0f + gamma[c] * (p[c] / denom);
}

// GRN statistics from the per-16-row column sums of squares the GEMM epilogue already produced
// (Epilogue::sumsq_out): one workgroup per sample sums the sample's row groups in fixed order, takes the channel
// mean and writes scale[b][c] = 1 + gamma[c] * Gx / (mean + 1e-6).  Replaces the two-pass kernels above on the
// per-step path (no re-read of the 4c-wide hidden tensor).
template <int G>  // G > 0: compile-time group count (all loads of a channel quad in flight together); 0 = runtime
__global__ __launch_bounds__(1024) void grn_from_partials_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                                 float* __restrict__ scale, int groups_rt, int C) {
    // grid (ceil(C/4096), B), 1024 threads: every workgroup recomputes the sample's channel mean (tiny, L2-resident, identical
    // fixed-order arithmetic in every workgroup -> deterministic) and then writes the scale of its own 4096 channels.  The launch
    // sits between the two GEMMs of every MLP block (256 times per image at batch 1), so it is built for latency: at most two
    // dependent load rounds per thread (C <= 8192), one shuffle tree, one barrier.
    __shared__ float red[16];
    const int groups = G > 0 ? G : groups_rt;
    const int b = blockIdx.y;
    const int C4 = C >> 2;
    const float* p = part + (size_t)b * groups * C;
    auto colsum = [&](int c4) {
        f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
        if (G > 0) {
            f32x4 v[G > 0 ? G : 1];
#pragma unroll
            for (int g = 0; g < G; ++g) v[g] = ld4(p + (size_t)g * C + c4 * 4);
#pragma unroll
            for (int g = 0; g < G; ++g) q += v[g];
        } else {
            for (int g = 0; g < groups; ++g) q += ld4(p + (size_t)g * C + c4 * 4);
        }
        return q;
    };
    float s = 0.f;
    f32x4 mine = f32x4{0.f, 0.f, 0.f, 0.f};
    const int my_c4 = blockIdx.x * 1024 + threadIdx.x;
    for (int c4 = threadIdx.x; c4 < C4; c4 += 1024) {
        const f32x4 q = colsum(c4);
        if (c4 == my_c4) mine = q;
        s += (sqrtf(q[0]) + sqrtf(q[1])) + (sqrtf(q[2]) + sqrtf(q[3]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];  // fixed wave order
    const float denom = tot / (float)C + 1e-6f;
    if (my_c4 < C4) {
        const f32x4 gm = ld4(gamma + my_c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = 1.0f + gm[e] * (sqrtf(mine[e]) / denom);
        st4(scale + (size_t)b * C + my_c4 * 4, o);
    }
}

int launch_grn_from_partials(const float* part, const float* gamma, float* scale, int B, int groups, int C, hipStream_t st) {
    if (B <= 0) return PAELLA_OK;
    if (C & 3) { paella_set_error("grn: C %% 4 != 0"); return PAELLA_ERR_ARG; }
    const dim3 grid((C / 4 + 1023) / 1024, B), block(1024);
    switch (groups) {
        case 1: hipLaunchKernelGGL((grn_from_partials_kernel<1>), grid, block, 0, st, part, gamma, scale, groups, C); break;
        case 4: hipLaunchKernelGGL((grn_from_partials_kernel<4>), grid, block, 0, st, part, gamma, scale, groups, C); break;
        case 16: hipLaunchKernelGGL((grn_from_partials_kernel<16>), grid, block, 0, st, part, gamma, scale, groups, C); break;
        default: hipLaunchKernelGGL((grn_from_partials_kernel<0>), grid, block, 0, st, part, gamma, scale, groups, C); break;
    }
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

int launch_grn_scale(const float* g, const float* gamma, float* scale, float* tmp_gx, int B, int rows_per_sample,
                     int C, hipStream_t st) {
    if (B <= 0) return PAELLA_OK;
    hipLaunchKernelGGL(grn_sumsq_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, g, tmp_gx, rows_per_sample, C);
    LAUNCH_CHECK_RET();
    hipLaunchKernelGGL(grn_finalize_kernel, dim3(B), dim3(256), 0, st, tmp_gx, gamma, scale, C);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// Token embedding + LayerNorm + PixelUnshuffle (reference src/modules.py:126-131,271):
// one wave per output position (= patch x patch tokens); output channel = c*p*p + dy*p + dx.
// ---------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void embed_ln_unshuffle_kernel(const int64_t* __restrict__ tokens,
                                                                                 const float* __restrict__ table,
                                                                                 float* __restrict__ out, int B, int H,
                                                                                 int W, int c_in, int num_labels, float eps, FastDiv dWo, FastDiv dHo) {
    const int lane = threadIdx.x & 63;
    const int Ho = H / P, Wo = W / P;
    const int64_t opos = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t total = (int64_t)B * Ho * Wo;
    if (opos >= total) return;
    // (output position -> (b, oy, ox) by multiply-high; total < 2^31 is checked by the launcher: three 64-bit divisions by runtime values were ~500 instructions
    // in front of the first token load)
    const unsigned orow_i = fast_div((unsigned)opos, dWo);  // b * Ho + oy
    const int ox = (int)((unsigned)opos - orow_i * (unsigned)Wo);
    const int64_t b = fast_div(orow_i, dHo);
    const int oy = (int)(orow_i - (unsigned)b * (unsigned)Ho);
    const int C4 = c_in >> 2;
    float* orow = out + opos * ((int64_t)c_in * P * P);
    // statistics per token (full row; the 8 MB table is L2/MALL resident), then the normalised
    // values are written channel-interleaved so the p*p tokens of a patch fill contiguous floats
#pragma unroll
    for (int t = 0; t < P * P; ++t) {
        const int dy = t / P, dx = t % P;
        int64_t tok = tokens[(b * H + (oy * P + dy)) * W + (ox * P + dx)];
        tok = tok < 0 ? 0 : (tok >= num_labels ? num_labels - 1 : tok);
        const float* row = table + tok * c_in;
        float s = 0.f;
        for (int c4 = lane; c4 < C4; c4 += 64) s += hsum4(ld4(row + c4 * 4));
        const float mean = wave_sum(s) / (float)c_in;
        float q = 0.f;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            f32x4 d = ld4(row + c4 * 4) - mean;
            d = d * d;
            q += hsum4(d);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)c_in + eps);
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const f32x4 n = (ld4(row + c4 * 4) - mean) * rstd;
#pragma unroll
            for (int e = 0; e < 4; ++e) orow[(c4 * 4 + e) * (P * P) + t] = n[e];
        }
    }
}

int launch_embed_ln_unshuffle(const int64_t* tokens, const float* table, float* out, int B, int H, int W, int c_in,
                              int patch, int num_labels, float eps, hipStream_t st) {
    if (c_in & 3) { paella_set_error("embed: c_in %% 4 != 0"); return PAELLA_ERR_ARG; }
    if (patch < 1 || patch > 2 || (H % patch) || (W % patch)) {
        paella_set_error("embed: unsupported patch_size %d for grid %dx%d (supported: 1, 2)", patch, H, W);
        return PAELLA_ERR_ARG;
    }
    const int64_t total = (int64_t)B * (H / patch) * (W / patch);
    if (total <= 0) return PAELLA_OK;
    if (total > 0x7fffffff) { paella_set_error("embed: too many positions"); return PAELLA_ERR_ARG; }
    const unsigned blocks = (unsigned)((total + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    const FastDiv dWo = fast_div_of((unsigned)(W / patch)), dHo = fast_div_of((unsigned)(H / patch));
    if (patch == 2)
        hipLaunchKernelGGL((embed_ln_unshuffle_kernel<2>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, st, tokens, table, out,
                           B, H, W, c_in, num_labels, eps, dWo, dHo);
    else
        hipLaunchKernelGGL((embed_ln_unshuffle_kernel<1>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, st, tokens, table, out,
                           B, H, W, c_in, num_labels, eps, dWo, dHo);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// Timestep embedding (reference src/modules.py:212-221) + all TimestepBlock mappers (:99-106) in one launch.
// freqs[k] = exp(-k*log(max_positions)/(half-1)) is computed on the host with torch so the sin/cos
// arguments are bit-identical to the reference's.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void timestep_kernel(const float* __restrict__ r, const float* __restrict__ freqs,
                                                       const float* __restrict__ Wcat, const float* __restrict__ bcat,
                                                       float* __restrict__ ts, int c_r, int total, float max_positions,
                                                       float* __restrict__ r_embed_out, int reps, int n_distinct) {
    extern __shared__ float emb[];
    const int b = blockIdx.y;
    const int half = c_r >> 1;
    const float rr = r[b] * max_positions;
    for (int k = threadIdx.x; k < c_r; k += 256) {
        float v = 0.f;
        if (k < half) v = sinf(rr * freqs[k]);
        else if (k < 2 * half) v = cosf(rr * freqs[k - half]);
        emb[k] = v;
        if (r_embed_out && blockIdx.x == 0) r_embed_out[(size_t)b * c_r + k] = v;
    }
    __syncthreads();
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const float* w = Wcat + (size_t)o * c_r;
    float acc = 0.f;
    for (int k = 0; k < c_r; ++k) acc += emb[k] * w[k];
    const float out = acc + bcat[o];
    for (int rep = 0; rep < reps; ++rep) ts[((size_t)rep * n_distinct + b) * total + o] = out;  // samples b, b + n_distinct, ... share r
}

int launch_timestep(const float* r, const float* freqs, const float* Wcat, const float* bcat, float* ts, int B,
                    int c_r, int total, float max_positions, float* r_embed_out, int reps, hipStream_t st) {
    if (B <= 0) return PAELLA_OK;
    int gx = (total + 255) / 256;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(timestep_kernel, dim3(gx, B), dim3(256), c_r * sizeof(float), st, r, freqs, Wcat, bcat, ts, c_r,
                       total, max_positions, r_embed_out, reps < 1 ? 1 : reps, B);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

__global__ __launch_bounds__(256) void scale_shift_kernel(float* __restrict__ x, const float* __restrict__ ts, int ts_stride,
                                                          int64_t rows, int rows_per_sample, int C) {
    const int C4 = C >> 2;
    const int64_t total = rows * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C4;
        const int c = (int)(i - row * C4) * 4;
        const float* t = ts + (row / rows_per_sample) * ts_stride;
        const f32x4 v = ld4(x + row * C + c);
        st4(x + row * C + c, v * (1.0f + ld4(t + c)) + ld4(t + C + c));
    }
}

int launch_scale_shift(float* x, const float* ts, int ts_stride, int64_t rows, int rows_per_sample, int C,
                       hipStream_t st) {
    if (rows <= 0) return PAELLA_OK;
    if (C & 3) { paella_set_error("scale_shift: C %% 4 != 0"); return PAELLA_ERR_ARG; }
    int64_t blocks = (rows * (C >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_shift_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ts, ts_stride, rows, rows_per_sample, C);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        y[i] = v / (1.0f + expf(-v));
    }
}
int launch_silu(const float* x, float* y, int64_t n, hipStream_t st) {
    if (n <= 0) return PAELLA_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, n);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd,
                                                        int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[r * lds_ + c];
    }
}
// the common case (everything a multiple of 4 floats, fewer than 2^31 quads): 16-byte copies, row index by multiply-high
__global__ __launch_bounds__(256) void copy_rows4_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, unsigned total4, int cols4, FastDiv dC) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const unsigned r = fast_div(i, dC);
        const unsigned c = (i - r * (unsigned)cols4) * 4u;
        st4(dst + (size_t)r * ldd + c, ld4(src + (size_t)r * lds_ + c));
    }
}
// x = a*x + b*y over n floats (n % 4 == 0): classifier-free-guidance mix of the two halves ahead of the linear head
__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ x, const float* __restrict__ y, float a, float b, int64_t n4, unsigned short* __restrict__ x16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = ld4(x + i * 4) * a + ld4(y + i * 4) * b;
        st4(x + i * 4, v);
        if (x16) st4h(x16 + i * 4, v);
    }
}
int launch_axpby(float* x, const float* y, float a, float b, int64_t n, hipStream_t st) { return launch_axpby16(x, y, a, b, n, nullptr, st); }
int launch_axpby16(float* x, const float* y, float a, float b, int64_t n, unsigned short* x16, hipStream_t st) {
    if (n <= 0) return PAELLA_OK;
    if (n & 3) { paella_set_error("axpby: n %% 4 != 0"); return PAELLA_ERR_ARG; }
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, a, b, n / 4, x16);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

int launch_copy_rows(const float* src, int lds_, float* dst, int ldd, int64_t rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return PAELLA_OK;
    if (!((cols | lds_ | ldd) & 3) && !(((uintptr_t)src | (uintptr_t)dst) & 15) && rows * (cols / 4) < 0x7fffffff) {
        const int64_t total4 = rows * (cols / 4);
        int64_t blocks4 = (total4 + 255) / 256;
        if (blocks4 > 4096) blocks4 = 4096;
        hipLaunchKernelGGL(copy_rows4_kernel, dim3((unsigned)blocks4), dim3(256), 0, st, src, lds_, dst, ldd, (unsigned)total4, cols / 4, fast_div_of((unsigned)(cols / 4)));
        LAUNCH_CHECK_RET();
        return PAELLA_OK;
    }
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// bf16 helpers of the opt-in fast mode (outside the fp32 parity contract)
// ---------------------------------------------------------------------------
// GlobalResponseNorm apply (reference src/modules.py:40: gamma * (x * Nx) + beta + x = x * scale[b][k] + beta[k]) IN PLACE on the bf16 hidden tensor of an
// MLP block: the second GEMM then takes a plain operand straight from HBM to LDS (a bf16 operand cannot be transformed on the way).  One 16-byte chunk
// (8 values) per thread and step; fp32 arithmetic, one rounding on the way back.
__global__ __launch_bounds__(256) void grn_apply16_kernel(unsigned short* __restrict__ h, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          unsigned total8, int C8, FastDiv dC8, FastDiv dRps) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total8; i += gridDim.x * 256u) {
        const unsigned row = fast_div(i, dC8);
        const unsigned c = (i - row * (unsigned)C8) * 8u;
        const unsigned b = fast_div(row, dRps);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(h + (size_t)i * 8);
        const float* sp = scale + (size_t)b * ((size_t)C8 * 8) + c;
        const f32x4 s0 = ld4(sp), s1 = ld4(sp + 4), t0 = ld4(shift + c), t1 = ld4(shift + c + 4);
        const f32x8 f = __builtin_convertvector(v, f32x8);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = f[e] * s0[e] + t0[e]; o[4 + e] = f[4 + e] * s1[e] + t1[e]; }
        *reinterpret_cast<bf16x8*>(h + (size_t)i * 8) = __builtin_convertvector(o, bf16x8);
    }
}
// The same for the throughput regime (round 6): a thread OWNS 8 columns of one sample and walks down 16 of its rows -- the scale / shift vectors are loaded once per
// thread instead of once per 16-byte chunk (the grid-stride form issues four side loads and a store per payload load: six vector-memory instructions per 16 bytes), and
// four payload loads are in flight per thread.  64 consecutive threads cover 1 KB of a row.  Identical arithmetic per element: bit-identical to the form above.
__global__ __launch_bounds__(256) void grn_apply16_cols_kernel(unsigned short* __restrict__ h, const float* __restrict__ scale, const float* __restrict__ shift, int C, int rows_per_sample) {
    const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;            // this thread's 8 columns
    const int b = blockIdx.z;                                            // sample
    const int r0 = blockIdx.y * 16 + (threadIdx.x >> 6);                 // rows r0, r0 + 4, r0 + 8, r0 + 12 of the sample
    unsigned short* hp = h + ((size_t)b * rows_per_sample + r0) * C + c;
    bf16x8 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const bf16x8*>(hp + (size_t)(4 * k) * C);
    const float* sp = scale + (size_t)b * C + c;
    const f32x4 s0 = ld4(sp), s1 = ld4(sp + 4), t0 = ld4(shift + c), t1 = ld4(shift + c + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x8 f = __builtin_convertvector(v[k], f32x8);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = f[e] * s0[e] + t0[e]; o[4 + e] = f[4 + e] * s1[e] + t1[e]; }
        *reinterpret_cast<bf16x8*>(hp + (size_t)(4 * k) * C) = __builtin_convertvector(o, bf16x8);
    }
}
int launch_grn_apply16(unsigned short* h, const float* scale, const float* shift, int64_t rows, int rows_per_sample, int C, hipStream_t st) {
    if (rows <= 0) return PAELLA_OK;
    if ((C & 7) || rows * (C / 8) >= 0x7fffffffll || rows_per_sample < 1) { paella_set_error("grn_apply16: C %% 8 != 0 or tensor too large"); return PAELLA_ERR_ARG; }
    if ((C & 511) == 0 && (rows_per_sample & 15) == 0 && rows % rows_per_sample == 0 && rows / rows_per_sample <= 65535 && rows_per_sample / 16 <= 65535) {
        hipLaunchKernelGGL(grn_apply16_cols_kernel, dim3((unsigned)(C / 512), (unsigned)(rows_per_sample / 16), (unsigned)(rows / rows_per_sample)), dim3(256), 0, st, h, scale, shift, C, rows_per_sample);
        LAUNCH_CHECK_RET();
        return PAELLA_OK;
    }
    const unsigned total8 = (unsigned)(rows * (C / 8));
    int64_t blocks = ((int64_t)total8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(grn_apply16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, h, scale, shift, total8, C / 8, fast_div_of((unsigned)(C / 8)), fast_div_of((unsigned)rows_per_sample));
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// Batch-1 regime of the same: statistics finalize (grn_from_partials_kernel) AND apply in ONE launch -- between the two MLP GEMMs of every block a launch
// costs ~4-5 us whatever it does.  grid (C / 256, B, rows per sample / 16): every workgroup re-derives the sample's channel mean of Gx from the producer's
// per-16-row partials (groups x C floats, L2-resident, fixed order -> every workgroup gets the same bits), computes the scale of its own 256 columns and applies
// it to its 16 rows (two rows per thread, both loads in flight).  Only for small launches (the redundant partial reads grow with the workgroup count).
__global__ __launch_bounds__(256) void grn_finalize_apply16_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ shift,
                                                                   unsigned short* __restrict__ h, int groups, int C, int rows_per_sample) {
    __shared__ float red[4];
    const int b = blockIdx.y, t = threadIdx.x;
    const int C4 = C >> 2;
    const float* p = part + (size_t)b * groups * C;
    const int c = blockIdx.x * 256 + (t & 31) * 8;  // this thread's 8 columns
    unsigned short* hb = h + ((size_t)b * rows_per_sample + blockIdx.z * 16 + (t >> 5)) * C + c;
    const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(hb), v1 = *reinterpret_cast<const bf16x8*>(hb + (size_t)8 * C);  // (issued first: they fly under the statistics)
    float s = 0.f;
    for (int c4 = t; c4 < C4; c4 += 256) {
        f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < groups; ++g) q += ld4(p + (size_t)g * C + c4 * 4);
        s += (sqrtf(q[0]) + sqrtf(q[1])) + (sqrtf(q[2]) + sqrtf(q[3]));
    }
    s = wave_sum(s);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    const float denom = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C + 1e-6f;
    f32x4 q0 = f32x4{0.f, 0.f, 0.f, 0.f}, q1 = q0;
    for (int g = 0; g < groups; ++g) { q0 += ld4(p + (size_t)g * C + c); q1 += ld4(p + (size_t)g * C + c + 4); }
    const f32x4 g0 = ld4(gamma + c), g1 = ld4(gamma + c + 4), t0 = ld4(shift + c), t1 = ld4(shift + c + 4);
    f32x4 s0, s1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0[e] = 1.0f + g0[e] * (sqrtf(q0[e]) / denom); s1[e] = 1.0f + g1[e] * (sqrtf(q1[e]) / denom); }
    const f32x8 f0 = __builtin_convertvector(v0, f32x8), f1 = __builtin_convertvector(v1, f32x8);
    f32x8 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o0[e] = f0[e] * s0[e] + t0[e]; o0[4 + e] = f0[4 + e] * s1[e] + t1[e];
        o1[e] = f1[e] * s0[e] + t0[e]; o1[4 + e] = f1[4 + e] * s1[e] + t1[e];
    }
    *reinterpret_cast<bf16x8*>(hb) = __builtin_convertvector(o0, bf16x8);
    *reinterpret_cast<bf16x8*>(hb + (size_t)8 * C) = __builtin_convertvector(o1, bf16x8);
}
// GlobalResponseNorm of the bf16 fast mode from the producing GEMM's per-16-row partials: scale (kept in `scale` for the two-kernel form) + apply in place
int launch_grn_partials_apply16(const float* part, const float* gamma, const float* shift, float* scale, unsigned short* h, int B, int rows_per_sample, int C, hipStream_t st) {
    if (B <= 0) return PAELLA_OK;
    if ((rows_per_sample & 15) || (C & 7)) { paella_set_error("grn_apply16: rows per sample %% 16 != 0 or C %% 8 != 0"); return PAELLA_ERR_ARG; }
    if ((C & 255) == 0 && (int64_t)B * rows_per_sample <= 2048) {
        hipLaunchKernelGGL(grn_finalize_apply16_kernel, dim3(C / 256, B, rows_per_sample / 16), dim3(256), 0, st, part, gamma, shift, h, rows_per_sample / 16, C, rows_per_sample);
        LAUNCH_CHECK_RET();
        return PAELLA_OK;
    }
    const int rc = launch_grn_from_partials(part, gamma, scale, B, rows_per_sample / 16, C, st);
    if (rc != PAELLA_OK) return rc;
    return launch_grn_apply16(h, scale, shift, (int64_t)B * rows_per_sample, rows_per_sample, C, st);
}

// fp32 -> bf16 (RNE): the shadow copy of a weight matrix, made once per (re)load
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4h(dst + i * 4, ld4(src + i * 4));
}
int launch_f32_to_bf16(const float* src, unsigned short* dst, size_t n, hipStream_t st) {
    if (n == 0) return PAELLA_OK;
    if (n & 3) { paella_set_error("f32_to_bf16: n %% 4 != 0"); return PAELLA_ERR_ARG; }
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n / 4);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
// out[n] = sum_k (float)W16[n][k]: the row sums a LayerNorm folded into a bf16 GEMM's epilogue needs -- of the ROUNDED weights, so that
// rstd * (sum_k a16 W16 - mean * wsum) is the LayerNorm of the rounded operand exactly.  One wave per row, fp32 sums in a fixed order.
__global__ __launch_bounds__(256) void rowsum_bf16_kernel(const unsigned short* __restrict__ W, float* __restrict__ out, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const unsigned short* w = W + (size_t)n * K;
    float s = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const f32x8 f = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(w + k), f32x8);
        s += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    }
    s = wave_sum(s);
    if (lane == 0) out[n] = s;
}
int launch_rowsum_bf16(const unsigned short* W, float* out, int N, int K, hipStream_t st) {
    if (N <= 0) return PAELLA_OK;
    if (K & 7) { paella_set_error("rowsum_bf16: K %% 8 != 0"); return PAELLA_ERR_ARG; }
    hipLaunchKernelGGL(rowsum_bf16_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, W, out, N, K);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// generic permute-copy (weight repack at load time; not on the per-step path)
// ---------------------------------------------------------------------------
struct PermuteArgs { int64_t oshape[5]; int64_t istride[5]; int ndim; int64_t total; };
__global__ __launch_bounds__(256) void permute_kernel(const float* __restrict__ src, float* __restrict__ dst, PermuteArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (int64_t)gridDim.x * 256) {
        int64_t rem = i, off = 0;
        for (int d = a.ndim - 1; d >= 0; --d) {
            const int64_t q = rem / a.oshape[d];
            off += (rem - q * a.oshape[d]) * a.istride[d];
            rem = q;
        }
        dst[i] = src[off];
    }
}
int launch_permute(const float* src, float* dst, const int64_t* shape, const int* perm, int ndim, hipStream_t st) {
    if (ndim < 1 || ndim > 5) { paella_set_error("permute: ndim %d unsupported", ndim); return PAELLA_ERR_ARG; }
    int64_t stride[5];
    int64_t s = 1;
    for (int d = ndim - 1; d >= 0; --d) { stride[d] = s; s *= shape[d]; }
    PermuteArgs a;
    a.ndim = ndim; a.total = s;
    for (int d = 0; d < ndim; ++d) { a.oshape[d] = shape[perm[d]]; a.istride[d] = stride[perm[d]]; }
    if (s <= 0) return PAELLA_OK;
    int64_t blocks = (s + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(permute_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, a);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
