// Device-side epilogue shared by the GEMM kernels (gemm.hip, gemm_bf16.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// bias -> activation -> alpha -> residual -> timestep scale/shift
__device__ __forceinline__ f32x4 epilogue_apply(const Epilogue& ep, int N, int m, int n, f32x4 v) {
    if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + n);
    if (ep.act == ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
    }
    if (ep.alpha != 1.0f) v *= ep.alpha;
    if (ep.residual) v += *reinterpret_cast<const f32x4*>(ep.residual + (size_t)m * ep.ldr + n);
    if (ep.ts) {
        const float* t = ep.ts + (size_t)(m / ep.rows_per_sample) * ep.ts_stride;
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
        *reinterpret_cast<f32x4*>(C + orow * ldc + n) = v;
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

__device__ __forceinline__ void epilogue_store(const Epilogue& ep, float* __restrict__ C, int ldc, int N,
                                               int m, int n, f32x4 v) {
    epilogue_write(ep, C, ldc, m, n, epilogue_apply(ep, N, m, n, v));
}
