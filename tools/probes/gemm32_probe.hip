// PROBE (not product code): how far does a plain fp32 GEMM get on MI355X with v_mfma_f32_32x32x2_f32 wave tiles, a 256x128 workgroup
// tile, direct-to-LDS operands and fragment reads pipelined one k-block ahead?  The vendor BLAS reaches 146-152 TFLOP/s on the
// largest shapes of the path (profiles/r02_vendor_blas_reference_probe.txt); the product kernel (16x16x4 MFMAs, 128x128 / 64x64 tiles)
// 133-140.  C[M,N] = A[M,K] . W[N,K]^T, M % 256 == 0, N % 128 == 0, K % 32 == 0, no epilogue fusion.
//
// Structure: 8 waves as 4 (M) x 2 (N), each a 64x64 wave tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  LDS: two stages of
// [256 + 128 rows][32 floats], 16-byte chunks XOR-swizzled by row % 8 on the SOURCE address (LDS-DMA writes lane-linearly).  A K step of
// 32 is four k-blocks of 8; a lane's ds_read_b128 of chunk 2b + lane/32 feeds the four MFMAs of block b (element e of both operands =
// the same two k's).  Fragments of block b + 1 are read while block b multiplies; the next K step's first block is read behind the
// barrier while the current step's last block multiplies.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/gemm32_probe.hip -o tools/probes/gemm32_probe.bin ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, NT = 512;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
#endif
}

template <int BM, int BN, int WM, int WN>  // workgroup tile, waves along M / N (WM * WN == 8); wave tile = (BM / WM) x (BN / WN) in 32x32 MFMA tiles
__global__ __launch_bounds__(NT) void gemm32_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int N, int K, int gm) {
    constexpr int TILE_FLOATS = (BM + BN) * BK;
    constexpr int TI = BM / WM / 32, TJ = BN / WN / 32, PA = BM / 64, PB = BN / 64;
    static_assert(WM * WN == 8, "8 waves");
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, h = lane >> 5;
    const int tiles_m = M / BM, tiles_n = N / BN;
    // XCD-aware order (workgroup b runs on XCD b % 8) + grouped rasterisation, as in the product kernel
    int t = blockIdx.x;
    {
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = t & 7, idx = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    if (gm >= tiles_m) { tile_m = t % tiles_m; tile_n = t / tiles_m; }
    else {
        const int width = gm * tiles_n, grp = t / width, rem = t - grp * width, first = grp * gm, gsz = min(tiles_m - first, gm);
        tile_n = rem / gsz;
        tile_m = first + (rem - tile_n * gsz);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)m0 * K), 0, BM * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (size_t)n0 * K), 0, BN * K * 4, 0x00020000);
    const int ldrow = tid >> 3, ldc = tid & 7;  // 64 rows per pass, 8 chunks of 16 bytes per row
    unsigned aoff[PA], boff[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) aoff[i] = ((unsigned)(ldrow + 64 * i) * (unsigned)K + (unsigned)((ldc ^ (ldrow & 7)) * 4)) * 4u;
#pragma unroll
    for (int i = 0; i < PB; ++i) boff[i] = ((unsigned)(ldrow + 64 * i) * (unsigned)K + (unsigned)((ldc ^ (ldrow & 7)) * 4)) * 4u;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    auto issue = [&](int stage, int kt) __attribute__((always_inline)) {
        float* dA = smem + stage * TILE_FLOATS + (wave * 8) * BK;
        float* dB = smem + stage * TILE_FLOATS + BM * BK + (wave * 8) * BK;
        const int kofs = kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < PA; ++i) dma16(rsrcA, dA + 64 * i * BK, aoff[i], kofs);
#pragma unroll
        for (int i = 0; i < PB; ++i) dma16(rsrcW, dB + 64 * i * BK, boff[i], kofs);
    };
    auto read_block = [&](int stage, int b, f32x4 (&af)[TI], f32x4 (&bf)[TJ]) __attribute__((always_inline)) {
        const float* As = smem + stage * TILE_FLOATS;
        const float* Bs = As + BM * BK;
        const int c = 2 * b + h;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int row = wm * (BM / WM) + i * 32 + r32;
            af[i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c ^ (row & 7)) << 2));
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int row = wn * (BN / WN) + j * 32 + r32;
            bf[j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + ((c ^ (row & 7)) << 2));
        }
    };
    auto mfma_block = [&](const f32x4 (&af)[TI], const f32x4 (&bf)[TJ]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);
    };

    const int KT = K / BK;
    f32x4 a0[TI], b0[TJ], a1[TI], b1[TJ];
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_block(0, 0, a0, b0);
    auto unit = [&](int s, int u) __attribute__((always_inline)) {
        if (u + 1 < KT) issue(s ^ 1, u + 1);
        __builtin_amdgcn_sched_barrier(0);
        read_block(s, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        read_block(s, 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        read_block(s, 3, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_block(s ^ 1, 0, a0, b0);  // next K step's first block (harmless re-read of stale data on the last step)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    };
    int u = 0;
    for (; u + 2 <= KT; u += 2) {
        unit(0, u);
        unit(1, u + 1);
    }
    if (u < KT) unit(0, u);

    // D = W_tile . A_tile^T: lane holds out[m = ..+r32][n = .. + 8g + 4h + 0..3] for g = 0..3
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 32 + r32;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * (BN / WN) + j * 32 + 8 * g + 4 * h;
                const f32x4 v = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(C + (size_t)m * N + n) = v;
            }
    }
}

__global__ void naive_kernel(const float* A, const float* W, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)m * K + k], W[(size_t)n * K + k], s);
    C[(size_t)m * N + n] = s;
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {  // uniform (-1, 1), full mantissas
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned s = (unsigned)i * 2654435761u + seed;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        p[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    }
}

static void fill(std::vector<float>& v, unsigned seed) {
    unsigned s = seed;
    for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
}

int main(int argc, char** argv) {
    // 1. correctness on a small problem with asymmetric data
    {
        const int M = 512, N = 256, K = 96;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N), hR((size_t)M * N);
        fill(hA, 1);
        fill(hW, 2);
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) hA[(size_t)m * K + k] += 0.01f * k + 0.001f * m;
        float *A, *W, *C, *R;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&C, hC.size() * 4); hipMalloc(&R, hR.size() * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
        hipMemset(C, 0xff, hC.size() * 4);
        hipLaunchKernelGGL((gemm32_kernel<256, 128, 4, 2>), dim3((M / 256) * (N / 128)), dim3(NT), 0, 0, A, W, C, M, N, K, 8);
        hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, W, R, M, N, K);
        hipDeviceSynchronize();
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mr = 0;
        for (size_t i = 0; i < hC.size(); ++i) { md = fmax(md, fabs((double)hC[i] - hR[i])); mr = fmax(mr, fabs((double)hR[i])); }
        printf("check %dx%dx%d: max |diff| vs naive fmaf kernel %.3e (max |ref| %.2f) -> %s\n", M, N, K, md, mr, md <= 1e-3 ? "OK" : "MISMATCH");
        hipFree(A); hipFree(W); hipFree(C); hipFree(R);
    }
    // 2. timing on the large shapes of the path (cold weights: rotate 3 copies)
    const int shapes[][3] = {{32768, 5120, 1280}, {32768, 1280, 5120}, {131072, 2560, 640}, {4096, 5120, 1280}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        float *A, *C, *W[3];
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, (size_t)M * K, 7u);
        { unsigned sd = 11u; for (auto& w : W) { hipMalloc(&w, (size_t)N * K * 4); hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, w, (size_t)N * K, sd++); } }
        auto run = [&](int variant, int gm, int i) {
            switch (variant) {
                case 0: hipLaunchKernelGGL((gemm32_kernel<256, 128, 4, 2>), dim3((M / 256) * (N / 128)), dim3(NT), 0, 0, A, W[i], C, M, N, K, gm); break;
                case 1: hipLaunchKernelGGL((gemm32_kernel<256, 256, 4, 2>), dim3((M / 256) * (N / 256)), dim3(NT), 0, 0, A, W[i], C, M, N, K, gm); break;
                case 2: hipLaunchKernelGGL((gemm32_kernel<256, 256, 2, 4>), dim3((M / 256) * (N / 256)), dim3(NT), 0, 0, A, W[i], C, M, N, K, gm); break;
                default: hipLaunchKernelGGL((gemm32_kernel<128, 256, 2, 4>), dim3((M / 128) * (N / 256)), dim3(NT), 0, 0, A, W[i], C, M, N, K, gm); break;
            }
        };
        const char* names[] = {"256x128 (4x2 waves)", "256x256 (4x2 waves)", "256x256 (2x4 waves)", "128x256 (2x4 waves)"};
        for (int variant = 0; variant < 4; ++variant)
        for (int gm : {8, 1 << 30}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) run(variant, gm, i);
            const int reps = M >= 32768 ? 9 : 30;
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) run(variant, gm, i % 3);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            printf("%6d x %5d x %5d  %s, 32x32x2 MFMA, DMA, %s: %8.1f us  %6.1f TFLOP/s\n", M, N, K, names[variant], gm == 8 ? "grouped raster 8" : "m-fastest order  ", us,
                   2.0 * M * N * K / us / 1e6);
        }
        hipFree(A); hipFree(C);
        for (auto& w : W) hipFree(w);
    }
    return 0;
}
