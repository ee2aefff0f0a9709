// PROBE (not product code): the batch-1 ("skinny") fp32 GEMMs of the sampling path with tiles that cover all 128 rows of M, 32x32x2 MFMA wave tiles,
// direct-to-LDS operands and split-K over grid.y (partials to a slab, summed by a second tiny launch here; the product combines in-launch).  Same main
// loop as tools/probes/gemm32_probe.hip.  The product kernel (32x32 workgroup tiles of four 16x16 waves, ~10 K-steps per workgroup) runs 128x5120x1280 in
// 23.0 us and 128x1280x5120 in 24.3 us; the vendor BLAS in 26.6 / 28.9 us; the matrix-core floor is 10.8 us.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/gemm32_skinny_probe.hip -o tools/probes/gemm32_skinny_probe.bin ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
#endif
}

template <int BM, int BN, int WM, int WN>  // workgroup tile, waves along M / N; wave tile = (BM / WM) x (BN / WN) in 32x32 MFMA tiles
// grid (tiles, S): workgroup (t, s) multiplies K slice s of tile t and writes its partial tile to slab[s][M][N] (S == 1: slab = C)
__global__ __launch_bounds__(64 * WM * WN) void gemm32_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int N, int K, int kslice) {
    constexpr int NT = 64 * WM * WN, RP = NT / 8;
    constexpr int TILE_FLOATS = (BM + BN) * BK;
    constexpr int TI = BM / WM / 32, TJ = BN / WN / 32, PA = BM / RP, PB = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0, "whole DMA passes");
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, h = lane >> 5;
    const int tiles_m = M / BM, tiles_n = N / BN;
    const int t = blockIdx.x;
    const int tile_m = t % tiles_m, tile_n = t / tiles_m;  // m fastest: neighbours share the weight panel
    const int kbeg = blockIdx.y * kslice;
    C += (size_t)blockIdx.y * M * N;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)m0 * K), 0, BM * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (size_t)n0 * K), 0, BN * K * 4, 0x00020000);
    const int ldrow = tid >> 3, ldc = tid & 7;  // RP rows per pass, 8 chunks of 16 bytes per row
    unsigned aoff[PA], boff[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) aoff[i] = ((unsigned)(ldrow + RP * i) * (unsigned)K + (unsigned)((ldc ^ (ldrow & 7)) * 4)) * 4u;
#pragma unroll
    for (int i = 0; i < PB; ++i) boff[i] = ((unsigned)(ldrow + RP * i) * (unsigned)K + (unsigned)((ldc ^ (ldrow & 7)) * 4)) * 4u;

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
        const int kofs = (kbeg + kt * BK) * 4;
#pragma unroll
        for (int i = 0; i < PA; ++i) dma16(rsrcA, dA + RP * i * BK, aoff[i], kofs);
#pragma unroll
        for (int i = 0; i < PB; ++i) dma16(rsrcW, dB + RP * i * BK, boff[i], kofs);
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

    const int KT = kslice / BK;
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


__global__ void reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int S, size_t mn4) {  // C = sum_s slab[s], float4 per thread
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= mn4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(slab)[i];
    for (int s = 1; s < S; ++s) v += reinterpret_cast<const f32x4*>(slab)[(size_t)s * mn4 + i];
    reinterpret_cast<f32x4*>(C)[i] = v;
}

__global__ void naive_kernel(const float* A, const float* W, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)m * K + k], W[(size_t)n * K + k], s);
    C[(size_t)m * N + n] = s;
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned s = (unsigned)i * 2654435761u + seed;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        p[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    }
}

template <int BM, int BN, int WM, int WN>
static void launch(const float* A, const float* W, float* slab, int M, int N, int K, int S) {
    hipLaunchKernelGGL((gemm32_kernel<BM, BN, WM, WN>), dim3((M / BM) * (N / BN), S), dim3(64 * WM * WN), 0, 0, A, W, slab, M, N, K, K / S);
}

int main() {
    const int shapes[][3] = {{128, 5120, 1280}, {128, 1280, 5120}, {512, 2560, 640}, {512, 640, 2560}, {128, 3840, 1280}, {128, 1280, 1280}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        const int ncopy = (int)(600e6 / ((double)N * K * 4)) + 1;
        float *A, *C, *R, *slab;
        std::vector<float*> W(ncopy);
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&R, (size_t)M * N * 4); hipMalloc(&slab, (size_t)64 * M * N * 4);
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, A, (size_t)M * K, 7u);
        for (int i = 0; i < ncopy; ++i) { hipMalloc(&W[i], (size_t)N * K * 4); hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, W[i], (size_t)N * K, 11u + i); }
        hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, W[0], R, M, N, K);
        std::vector<float> hC((size_t)M * N), hR((size_t)M * N);
        hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost);
        printf("%d x %d x %d (MFMA floor %.1f us at 155.6 TFLOP/s; weights %.1f MB; %d cold copies)\n", M, N, K, 2.0 * M * N * K / 155.6e6, N * K * 4e-6, ncopy);
        for (int variant = 0; variant < 3; ++variant) {
            const int bn = variant == 1 ? 32 : 64, units = K / 32;
            for (int S : {1, 2, 4, 5, 8, 10, 16, 20, 32, 40}) {
                if (units % S || units / S < 2) continue;
                const long wgs = (long)(M / 128) * (N / bn) * S;
                if (wgs < 200 || wgs > 2600) continue;
                auto go = [&](const float* w) {
                    if (variant == 0) launch<128, 64, 4, 2>(A, w, S == 1 ? C : slab, M, N, K, S);
                    else if (variant == 1) launch<128, 32, 4, 1>(A, w, S == 1 ? C : slab, M, N, K, S);
                    else launch<128, 64, 2, 2>(A, w, S == 1 ? C : slab, M, N, K, S);
                };
                auto red = [&]() { if (S > 1) hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)(((size_t)M * N / 4 + 255) / 256)), dim3(256), 0, 0, slab, C, S, (size_t)M * N / 4); };
                go(W[0]); red();
                hipDeviceSynchronize();
                hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
                double md = 0;
                for (size_t i = 0; i < hC.size(); ++i) md = fmax(md, fabs((double)hC[i] - hR[i]));
                float ms_g = 0, ms_gr = 0;
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {  // rep 0: GEMM launches only; rep 1: GEMM + reduce
                    hipEventRecord(e0);
                    for (int i = 0; i < ncopy; ++i) { go(W[i]); if (rep) red(); }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(rep ? &ms_gr : &ms_g, e0, e1);
                }
                const char* nm[] = {"128x64, 8 waves of 32x32", "128x32, 4 waves of 32x32", "128x64, 4 waves of 64x32"};
                printf("  tile %s, split-K %2d (%4ld workgroups): partial GEMM %6.1f us, + separate reduce launch %6.1f us   max|diff| %.1e %s\n", nm[variant], S, wgs,
                       ms_g * 1e3 / ncopy, ms_gr * 1e3 / ncopy, md, md <= 2e-3 ? "" : "MISMATCH");
            }
        }
        hipFree(A); hipFree(C); hipFree(R); hipFree(slab);
        for (auto w : W) hipFree(w);
    }
    return 0;
}
