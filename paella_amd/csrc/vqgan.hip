// VQGAN-side data-movement kernels for gfx950 (reference src/vqgan.py:54-107).  The dense contractions
// (1x1 convs, MLPs, the 4 phases of ConvTranspose2d(k4,s2,p1), Conv2d(k4,s2,p1) as im2col) run on the
// shared fp32 MFMA GEMM; these kernels only gather / scatter NHWC rows with 16-byte lanes.
#include "common.h"
#include <math.h>

__device__ __forceinline__ f32x4 ld4v(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4v(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// idx2vq (reference src/vqgan.py:104; torchtools VectorQuantize.idx2vq = embedding lookup)
__global__ __launch_bounds__(256) void codebook_gather_kernel(const int64_t* __restrict__ idx, const float* __restrict__ cb,
                                                              float* __restrict__ out, int64_t rows, int D, int K, float scale) {
    const int64_t total = rows * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        int64_t k = idx[r];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
        out[i] = cb[k * D + d] * scale;
    }
}
int launch_codebook_gather(const int64_t* idx, const float* codebook, float* out, int64_t rows, int D, int K, float scale,
                           hipStream_t st) {
    if (rows <= 0) return PAELLA_OK;
    int64_t blocks = (rows * D + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(codebook_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, codebook, out, rows, D, K, scale);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ConvTranspose2d(k=4, s=2, p=1) (reference src/vqgan.py:83-85) split into 4 output phases (py,px):
// out[b, 2y+py, 2x+px, co] = sum over 2x2 taps (ty,tx) and ci of x[b, y+oy(ty), x+ox(tx), ci] * W[ci, co, ky, kx]
// with (py=0: (oy,ky) in {(0,1),(-1,3)}; py=1: {(1,0),(0,2)}), same for x.  This kernel builds the phase's
// A operand [B*H*W, 4*C] (k index = (ty*2+tx)*C + ci), zero where the tap falls outside the input.
__global__ __launch_bounds__(256) void convT4_gather_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H,
                                                            int W, int C, int py, int px) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * H * W * 4 * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t rem = i / C4;
        const int tap = (int)(rem & 3);
        const int64_t pos = rem >> 2;
        const int xx = (int)(pos % W);
        const int yy = (int)((pos / W) % H);
        const int ty = tap >> 1, tx = tap & 1;
        const int oy = py == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 1 : 0);
        const int ox = px == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 1 : 0);
        const int sy = yy + oy, sx = xx + ox;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = ld4v(x + (pos + (int64_t)oy * W + ox) * C + c4 * 4);
        st4v(out + pos * (4 * (int64_t)C) + tap * C + c4 * 4, v);
    }
}
int launch_convT4_gather(const float* x, float* out, int B, int H, int W, int C, int py, int px, hipStream_t st) {
    if (C & 3) { paella_set_error("convT4_gather: C %% 4 != 0"); return PAELLA_ERR_ARG; }
    const int64_t total = (int64_t)B * H * W * C;
    if (total <= 0) return PAELLA_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(convT4_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, out, B, H, W, C, py, px);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// Conv2d(k=4, s=2, p=1) (reference src/vqgan.py:61) as im2col: out[(b,oy,ox)][(ky*4+kx)*C + c] = x[b, 2oy-1+ky, 2ox-1+kx, c]
__global__ __launch_bounds__(256) void conv4s2_im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H,
                                                             int W, int C) {
    const int C4 = C >> 2;
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)B * Ho * Wo * 16 * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t rem = i / C4;
        const int tap = (int)(rem & 15);
        const int64_t opos = rem >> 4;
        const int ox = (int)(opos % Wo);
        const int oy = (int)((opos / Wo) % Ho);
        const int64_t b = opos / ((int64_t)Ho * Wo);
        const int sy = 2 * oy - 1 + (tap >> 2), sx = 2 * ox - 1 + (tap & 3);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = ld4v(x + ((b * H + sy) * W + sx) * C + c4 * 4);
        st4v(out + opos * (16 * (int64_t)C) + tap * C + c4 * 4, v);
    }
}
int launch_conv4s2_im2col(const float* x, float* out, int B, int H, int W, int C, hipStream_t st) {
    if ((C & 3) || (H & 1) || (W & 1)) { paella_set_error("conv4s2_im2col: bad shape"); return PAELLA_ERR_ARG; }
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * 4 * C;
    if (total <= 0) return PAELLA_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(conv4s2_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, out, B, H, W, C);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// PixelUnshuffle(2) of an NCHW image into NHWC rows (reference src/vqgan.py:55): out[(b,y,x)][c*4+dy*2+dx] = img[b][c][2y+dy][2x+dx]
__global__ __launch_bounds__(256) void img_unshuffle_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int C,
                                                            int Hp, int Wp) {
    const int Ho = Hp >> 1, Wo = Wp >> 1, Co = C * 4;
    const int64_t total = (int64_t)B * Ho * Wo * Co;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % Co);
        const int64_t pos = i / Co;
        const int x = (int)(pos % Wo);
        const int y = (int)((pos / Wo) % Ho);
        const int64_t b = pos / ((int64_t)Ho * Wo);
        const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
        out[i] = img[((b * C + c) * Hp + 2 * y + dy) * Wp + 2 * x + dx];
    }
}
int launch_img_unshuffle(const float* img, float* out, int B, int C, int Hp, int Wp, hipStream_t st) {
    if ((Hp & 1) || (Wp & 1)) { paella_set_error("img_unshuffle: odd image size"); return PAELLA_ERR_ARG; }
    const int64_t total = (int64_t)B * C * Hp * Wp;
    if (total <= 0) return PAELLA_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(img_unshuffle_kernel, dim3((unsigned)blocks), dim3(256), 0, st, img, out, B, C, Hp, Wp);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// Nearest codebook row.  torchtools.nn.VectorQuantize (pabloppp/pytorch-tools, unpinned in
// reference requirements.txt:12) computes dist = (|e|^2 + |x|^2) - 2 x.e^T and takes min(dim=1);
// first minimum wins.  One wave per row, lanes stride the codebook, (min, index) shuffle reduction.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                         int64_t* __restrict__ idx, float* __restrict__ qe, int64_t rows, int D,
                                                         int K) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float xsq = 0.f;
    for (int d = 0; d < D; ++d) xsq = __fadd_rn(xsq, __fmul_rn(xr[d], xr[d]));
    float best = INFINITY;
    int best_k = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        const float* e = cb + (size_t)k * D;
        float esq = 0.f, dot = 0.f;
        for (int d = 0; d < D; ++d) {
            esq = __fadd_rn(esq, __fmul_rn(e[d], e[d]));
            dot = fmaf(xr[d], e[d], dot);
        }
        const float dist = __fadd_rn(__fadd_rn(esq, xsq), __fmul_rn(-2.0f, dot));
        if (dist < best || (dist == best && k < best_k)) { best = dist; best_k = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int ok = __shfl_xor(best_k, o, 64);
        if (ov < best || (ov == best && ok < best_k)) { best = ov; best_k = ok; }
    }
    if (best_k == 0x7fffffff) best_k = 0;
    if (lane == 0) idx[row] = best_k;
    if (qe && lane < D) qe[row * D + lane] = cb[(size_t)best_k * D + lane];
}
// The same search for the shipped geometry (c_latent = 4, codebook <= 8192 rows: 128 KiB) with the codebook RESIDENT IN LDS: one 512-thread workgroup per CU
// stages it once and walks 64-row chunks (8 waves x 8 rows); a lane holds the latents of its wave's 8 rows in registers, reads each of its codes
// (k = lane, lane + 64, ...) from LDS once per chunk and reuses it -- and its |e|^2 -- for the 8 rows.  The wave-per-row kernel above re-fetches the whole
// codebook per row through L1 / L2 (34 GB for the 262 144 rows of a 1024 px batch of 16: 7.9 ms).  The arithmetic per (row, code) pair is the
// SAME sequence of fp32 operations and the same first-minimum tie-break, so the indices are bit-identical to the kernel above.
__global__ __launch_bounds__(512) void vq_nearest_lds_kernel(const float* __restrict__ x, const float* __restrict__ cb, int64_t* __restrict__ idx,
                                                             float* __restrict__ qe, int64_t rows, int K) {
    constexpr int D = 4, R = 8, KMAX = 8192;
    __shared__ __attribute__((aligned(16))) float cbs[KMAX * D];
    for (int i = threadIdx.x; i < K; i += 512) *reinterpret_cast<f32x4*>(cbs + i * D) = *reinterpret_cast<const f32x4*>(cb + (size_t)i * D);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nchunks = (rows + 8 * R - 1) / (8 * R);
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t row0 = chunk * (8 * R) + wave * R;
        f32x4 xr[R];
        float xsq[R], best[R];
        int best_k[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r < rows ? row0 + r : rows - 1;  // clamp: out-of-range rows compute and are not stored
            xr[r] = *reinterpret_cast<const f32x4*>(x + row * D);
            float q = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) q = __fadd_rn(q, __fmul_rn(xr[r][d], xr[r][d]));
            xsq[r] = q;
            best[r] = INFINITY;
            best_k[r] = 0x7fffffff;
        }
        for (int k = lane; k < K; k += 64) {
            const f32x4 e = *reinterpret_cast<const f32x4*>(cbs + k * D);
            float esq = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) esq = __fadd_rn(esq, __fmul_rn(e[d], e[d]));
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float dot = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) dot = fmaf(xr[r][d], e[d], dot);
                const float dist = __fadd_rn(__fadd_rn(esq, xsq[r]), __fmul_rn(-2.0f, dot));
                if (dist < best[r]) { best[r] = dist; best_k[r] = k; }  // k increases per lane: a strict < keeps the first minimum
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float b = best[r];
            int bk = best_k[r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(b, o, 64);
                const int ok = __shfl_xor(bk, o, 64);
                if (ov < b || (ov == b && ok < bk)) { b = ov; bk = ok; }
            }
            if (bk == 0x7fffffff) bk = 0;
            const int64_t row = row0 + r;
            if (row < rows) {
                if (lane == 0) idx[row] = bk;
                if (qe && lane < D) qe[row * D + lane] = cbs[bk * D + lane];
            }
        }
    }
}

int launch_vq_nearest(const float* x, const float* codebook, int64_t* idx, float* qe, int64_t rows, int D, int K,
                      hipStream_t st) {
    if (rows <= 0) return PAELLA_OK;
    if (D > 64) { paella_set_error("vq_nearest: latent dim %d > 64 unsupported", D); return PAELLA_ERR_ARG; }
    if (D == 4 && K <= 8192 && rows >= 4096) {  // enough rows to amortise staging the codebook per workgroup
        const int64_t nchunks = (rows + 63) / 64;
        const unsigned blocks = (unsigned)(nchunks < 256 ? nchunks : 256);
        hipLaunchKernelGGL(vq_nearest_lds_kernel, dim3(blocks), dim3(512), 0, st, x, codebook, idx, qe, rows, K);
        LAUNCH_CHECK_RET();
        return PAELLA_OK;
    }
    hipLaunchKernelGGL(vq_nearest_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, codebook, idx, qe, rows, D, K);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

__global__ __launch_bounds__(256) void affine_cols_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ y, int64_t total, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        y[i] = __fadd_rn(__fmul_rn(x[i], scale[c]), shift[c]);
    }
}
int launch_affine_cols(const float* x, const float* scale, const float* shift, float* y, int64_t rows, int C, hipStream_t st) {
    const int64_t total = rows * C;
    if (total <= 0) return PAELLA_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(affine_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, scale, shift, y, total, C);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C,
                                                           float scale, int divide) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i indexes the NCHW output
        const int p = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const int64_t b = i / ((int64_t)HW * C);
        const float v = x[(b * HW + p) * C + c];
        y[i] = divide ? __fdiv_rn(v, scale) : __fmul_rn(v, scale);
    }
}
int launch_nhwc_to_nchw(const float* x, float* y, int B, int HW, int C, float scale, int divide, hipStream_t st) {
    const int64_t total = (int64_t)B * HW * C;
    if (total <= 0) return PAELLA_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, B, HW, C, scale, divide);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C,
                                                           float scale, int divide) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i indexes the NHWC output
        const int c = (int)(i % C);
        const int p = (int)((i / C) % HW);
        const int64_t b = i / ((int64_t)HW * C);
        const float v = x[(b * C + c) * HW + p];
        y[i] = divide ? __fdiv_rn(v, scale) : __fmul_rn(v, scale);
    }
}
int launch_nchw_to_nhwc(const float* x, float* y, int B, int HW, int C, float scale, int divide, hipStream_t st) {
    const int64_t total = (int64_t)B * HW * C;
    if (total <= 0) return PAELLA_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, B, HW, C, scale, divide);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
