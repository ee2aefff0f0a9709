// Sustained fp32 matrix-core peak of the device: nothing but independent v_mfma_f32_16x16x4_f32 chains, every SIMD of every CU,
// for ~10 ms at a time.  Prints TFLOP/s, the implied shader clock (64 flop / cycle / SIMD) and the s_memtime-measured clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak_probe.hip -o tools/probes/mfma_peak_probe.bin ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// RANDOM = false: the same two operand values in every MFMA (minimal switching activity); true: 8 + 8 different random operand
// registers per lane, cycled, so consecutive MFMAs see unrelated mantissas like a real GEMM's
template <bool RANDOM>
__global__ __launch_bounds__(256) void mfma_only(float* out, int iters, unsigned long long* clk, const float* rnd) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        av[i] = RANDOM ? rnd[(threadIdx.x * 16 + i) & 4095] : (float)(threadIdx.x & 7) * 0.125f;
        bv[i] = RANDOM ? rnd[(threadIdx.x * 16 + 8 + i) & 4095] : (float)(threadIdx.x & 3) * 0.25f;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[(i + r) & 7], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
    const bool random_ops = argc > 2 && atoi(argv[2]) != 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one wave per SIMD per block
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    float* rnd;
    hipMalloc(&rnd, 4096 * sizeof(float));
    {
        float h[4096];
        unsigned s = 12345u;
        for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }  // uniform (-1, 1), full mantissas
        hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device: %s, %d CUs, nominal clock %d MHz, %d wave(s) per SIMD, wall clock %d kHz, operands: %s\n", prop.name, cus, prop.clockRate / 1000, waves_per_simd, wall_khz,
           random_ops ? "8+8 random registers per lane, cycled" : "constant");
    for (int rep = 0; rep < 6; ++rep) {
        const int iters = rep < 2 ? 20000 : 200000;  // ~1 ms, then ~10 ms launches
        hipEventRecord(e0);
        if (random_ops) hipLaunchKernelGGL(mfma_only<true>, dim3(blocks), dim3(256), 0, 0, out, iters, clk, rnd);
        else hipLaunchKernelGGL(mfma_only<false>, dim3(blocks), dim3(256), 0, 0, out, iters, clk, rnd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double mfmas = (double)blocks * 4 * (double)iters * 32;
        const double tf = mfmas * 2048.0 / (ms * 1e-3) / 1e12;
        const double clock_from_rate = mfmas / ((double)cus * 4) * 32.0 / (ms * 1e-3) / 1e9;  // MFMA issue cycles per SIMD / time
        const double shader_ghz = wall_khz > 0 ? (double)h[0] / ((double)h[1] / (wall_khz * 1e3)) / 1e9 : 0.0;
        printf("launch %d: %.3f ms, %.1f TFLOP/s fp32 MFMA (16x16x4), implied clock if 32 cycles per MFMA per SIMD: %.3f GHz, cycle counter / wall clock: %.3f GHz\n", rep, ms, tf, clock_from_rate, shader_ghz);
    }
    return 0;
}
