// UNet ResBlock front half for gfx950 (reference src/modules.py:46-47,55-58):
//   depthwise Conv2d(k=3, zero padding, groups=C) + bias  ->  LayerNorm2d(C, no affine, eps 1e-6)
// NHWC fp32.  One 256-thread workgroup per output position; thread t owns 16-byte channel slots t, t+256, ...
// so every tap is one fully coalesced row read.  Latency matters more than bandwidth here (batch-1 sampling runs
// this on 32..512 positions): all 9 x NV activation loads and 9 x NV weight loads of a thread are issued
// unconditionally from clamped addresses (zero padding = a 0/1 factor on the tap, not a branch) so they are all in
// flight together, and the LayerNorm statistics take two block reductions (two-pass mean / variance like torch).
// Skip variant = Conv2d(2C -> C, groups=C) over cat([x, skip]): output channel g reads concatenated channels 2g, 2g+1;
// weights repacked [j][tap][C].
#include "common.h"

__device__ __forceinline__ f32x4 ldq(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

template <int NV, bool SKIP>
__global__ __launch_bounds__(256) void dwconv_ln_block_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ y, unsigned short* __restrict__ y16, int H, int W, int C, float eps, FastDiv dW, FastDiv dH) {
    __shared__ float red[4];
    const int64_t pos = blockIdx.x;
    const int C4 = C >> 2;
    // (position -> (y, x) by multiply-high: a 64-bit division by a runtime value is ~100 instructions in front of this latency-bound kernel's first load)
    const unsigned prow = fast_div((unsigned)blockIdx.x, dW);  // b * H + y
    const int xx = (int)((unsigned)blockIdx.x - prow * (unsigned)W);
    const int yy = (int)(prow - fast_div(prow, dH) * (unsigned)H);
    const int64_t img = pos - (int64_t)yy * W - xx;  // row index of (b, 0, 0)
    int c4s[NV];
    bool live[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = threadIdx.x + i * 256;
        live[i] = c4 < C4;
        c4s[i] = live[i] ? c4 : C4 - 1;
    }
    f32x4 acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = ldq(bias + c4s[i] * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int sy = yy + ky - 1;
        const int syc = min(max(sy, 0), H - 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sx = xx + kx - 1;
            const int sxc = min(max(sx, 0), W - 1);
            const float keep = (sy == syc && sx == sxc) ? 1.0f : 0.0f;  // zero padding
            const int64_t npos = img + (int64_t)syc * W + sxc;
            const int tap = ky * 3 + kx;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (!SKIP) {
                    acc[i] += ldq(x + npos * C + c4s[i] * 4) * (ldq(w + tap * C + c4s[i] * 4) * keep);
                } else {
                    const int cc = 8 * c4s[i];  // out channels 4*c4..+3 read cat channels 2g..2g+7
                    const float* src = cc < C ? (x + npos * C + cc) : (skip + npos * C + (cc - C));
                    const f32x4 e0 = ldq(src), e1 = ldq(src + 4);
                    const f32x4 w0 = ldq(w + tap * C + c4s[i] * 4) * keep;
                    const f32x4 w1 = ldq(w + (9 + tap) * C + c4s[i] * 4) * keep;
                    acc[i][0] += e0[0] * w0[0] + e0[1] * w1[0];
                    acc[i][1] += e0[2] * w0[1] + e0[3] * w1[1];
                    acc[i][2] += e1[0] * w0[2] + e1[1] * w1[2];
                    acc[i][3] += e1[2] * w0[3] + e1[3] * w1[3];
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (live[i]) s += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    const float mean = block_sum_256(s, red) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (live[i]) {
            f32x4 d = acc[i] - mean;
            d = d * d;
            q += (d[0] + d[1]) + (d[2] + d[3]);
        }
    const float rstd = 1.0f / sqrtf(block_sum_256(q, red) / (float)C + eps);
    if (y16) {  // kernel-uniform: bf16 output for a consuming bf16 GEMM (opt-in fast mode; y may be null then)
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (live[i]) *reinterpret_cast<bf16x4*>(y16 + pos * C + (threadIdx.x + i * 256) * 4) = __builtin_convertvector((acc[i] - mean) * rstd, bf16x4);
    }
    if (y) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (live[i]) *reinterpret_cast<f32x4*>(y + pos * C + (threadIdx.x + i * 256) * 4) = (acc[i] - mean) * rstd;
    }
}

int launch_dwconv_ln(const float* x, const float* skip, const float* w, const float* bias, float* y, int B, int H,
                     int W, int C, float eps, hipStream_t st, unsigned short* y16) {
    const int64_t total = (int64_t)B * H * W;
    if (total <= 0) return PAELLA_OK;
    // (the skip variant holds two activation loads per tap: its 8-slot instantiation spilled 1.9 KB per lane to scratch -- it stops at 4096 channels, 3x the widest released level)
    if ((C & 3) || (skip && (C & 7)) || C > 8192 || (skip && C > 4096)) { paella_set_error("dwconv_ln: bad channel count %d", C); return PAELLA_ERR_ARG; }
    if (total > 0x7fffffff) { paella_set_error("dwconv_ln: too many positions"); return PAELLA_ERR_ARG; }
    const int nv = (C / 4 + 255) / 256;
    const dim3 grid((unsigned)total), block(256);
    const FastDiv dW = fast_div_of((unsigned)W), dH = fast_div_of((unsigned)H);
#define DW_LAUNCH(NVv)                                                                                                   \
    do {                                                                                                                 \
        if (skip) hipLaunchKernelGGL((dwconv_ln_block_kernel<NVv, true>), grid, block, 0, st, x, skip, w, bias, y, y16, H, W, C, eps, dW, dH); \
        else hipLaunchKernelGGL((dwconv_ln_block_kernel<NVv, false>), grid, block, 0, st, x, skip, w, bias, y, y16, H, W, C, eps, dW, dH);     \
    } while (0)
    if (nv <= 1) DW_LAUNCH(1);
    else if (nv <= 2) DW_LAUNCH(2);
    else if (nv <= 4) DW_LAUNCH(4);
    else hipLaunchKernelGGL((dwconv_ln_block_kernel<8, false>), grid, block, 0, st, x, skip, w, bias, y, y16, H, W, C, eps, dW, dH);
#undef DW_LAUNCH
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
