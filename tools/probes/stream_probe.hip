// Weight-stream probe: how fast can N x K fp32 be pulled from HBM with the GEMM's own addressing (32-row x 32-float
// k-tiles, one float4 per thread per k-tile) as a function of the number of k-tiles kept in flight per workgroup?
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe.hip -o tools/probes/stream_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int D, bool TILED>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ W, float* __restrict__ out, int K, int kslice) {
    const int t = threadIdx.x, nt = blockIdx.x, ks = blockIdx.y;
    const int row = t >> 3, c4 = t & 7;
    const int ktiles = kslice / 32;
    const float4* base;
    size_t step;
    if (TILED) {  // [N/32][K/32][32x32] contiguous 4 KB tiles
        base = reinterpret_cast<const float4*>(W + ((size_t)nt * (K / 32) + (size_t)ks * ktiles) * 1024) + t;
        step = 256;
    } else {
        base = reinterpret_cast<const float4*>(W + (size_t)(nt * 32 + row) * K + (size_t)ks * kslice) + c4;
        step = 8;
    }
    float acc = 0.f;
    for (int k = 0; k < ktiles; k += D) {
        float4 r[D];
#pragma unroll
        for (int d = 0; d < D; ++d) r[d] = base[(size_t)(k + d) * step];
#pragma unroll
        for (int d = 0; d < D; ++d) acc += r[d].x + r[d].y + r[d].z + r[d].w;
    }
    if (acc == 12345.678f) out[t] = acc;
}

// MFMA-fragment addressing: each wave owns 16 rows; lane (r16 = lane&15, kq = lane>>4) reads 16 B at k0 + 4*kq, i.e. a
// wave instruction touches 16 rows x 64 B; consecutive instructions advance 64 B.  NG loads in flight per lane.
template <int NG>
__global__ __launch_bounds__(256) void probe_frag(const float* __restrict__ W, float* __restrict__ out, int K, int kslice) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const float* row = W + (size_t)(blockIdx.x * 64 + wave * 16 + r16) * K + (size_t)blockIdx.y * kslice + kq * 4;
    float acc = 0.f;
    for (int k = 0; k < kslice; k += NG * 16) {
        float4 r[NG];
#pragma unroll
        for (int d = 0; d < NG; ++d) r[d] = *reinterpret_cast<const float4*>(row + k + d * 16);
#pragma unroll
        for (int d = 0; d < NG; ++d) acc += r[d].x + r[d].y + r[d].z + r[d].w;
    }
    if (acc == 12345.678f) out[t] = acc;
}

template <int NG>
float run_frag(const std::vector<float*>& Ws, float* out, int N, int K, int S, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid(N / 64, S);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, st);
        for (float* W : Ws) probe_frag<NG><<<grid, 256, 0, st>>>(W, out, K, K / S);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / Ws.size();
}

// Operand-fetch phase of a "no LDS, no barrier" skinny GEMM: 32x32 output tile per 4-wave workgroup, the four waves split the
// K slice, every wave pulls its A (2 x 16 rows) and W (2 x 16 rows) MFMA fragments straight from global memory / L2.
// Measures whether the vector-memory path can deliver the 8 KB per 32x32x32 tile step faster than the LDS-staged kernel.
template <int NG>
__global__ __launch_bounds__(256) void probe_gemm_frag(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out,
                                                       int M, int N, int K, int kslice) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int tiles_m = M / 32;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int kq4 = kq * 4;
    const float* a0 = A + (size_t)(tm * 32 + r16) * K + (size_t)blockIdx.y * kslice + kq4;
    const float* a1 = a0 + (size_t)16 * K;
    const float* w0 = W + (size_t)(tn * 32 + r16) * K + (size_t)blockIdx.y * kslice + kq4;
    const float* w1 = w0 + (size_t)16 * K;
    float acc = 0.f;
    const int groups = kslice / 16;  // 16-k groups of the slice; wave w takes groups w, w+4, ...
    for (int g0 = wave; g0 < groups; g0 += 4 * NG) {
        float4 ra[NG], rb[NG], rc[NG], rd[NG];
#pragma unroll
        for (int d = 0; d < NG; ++d) {
            const int k = min(g0 + 4 * d, groups - 1) * 16;
            ra[d] = *reinterpret_cast<const float4*>(a0 + k);
            rb[d] = *reinterpret_cast<const float4*>(a1 + k);
            rc[d] = *reinterpret_cast<const float4*>(w0 + k);
            rd[d] = *reinterpret_cast<const float4*>(w1 + k);
        }
#pragma unroll
        for (int d = 0; d < NG; ++d) acc += ra[d].x + rb[d].y + rc[d].z + rd[d].w;
    }
    if (acc == 12345.678f) out[t] = acc;
}

template <int NG>
float run_gemm_frag(const float* A, const std::vector<float*>& Ws, float* out, int M, int N, int K, int S, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid((M / 32) * (N / 32), S);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, st);
        for (float* W : Ws) probe_gemm_frag<NG><<<grid, 256, 0, st>>>(A, W, out, M, N, K, K / S);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / Ws.size();
}

template <int D, bool TILED>
float run(const std::vector<float*>& Ws, float* out, int N, int K, int S, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid(N / 32, S);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, st);
        for (float* W : Ws) probe<D, TILED><<<grid, 256, 0, st>>>(W, out, K, K / S);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / Ws.size();
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 5120, K = argc > 2 ? atoi(argv[2]) : 1280;
    size_t bytes = (size_t)N * K * 4;
    int ncopy = (int)(700e6 / bytes) + 1;
    if (ncopy > 48) ncopy = 48;
    std::vector<float*> Ws(ncopy);
    for (auto& p : Ws) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
    float* out;
    hipMalloc(&out, 4096);
    hipStream_t st;
    hipStreamCreate(&st);
    printf("N=%d K=%d (%.1f MB per launch, %d rotating copies)\n", N, K, bytes / 1e6, ncopy);
    for (int S : {1, 2, 4, 8}) {
        if ((K / S) % 256) continue;
        float a1 = run<1, false>(Ws, out, N, K, S, st), a2 = run<2, false>(Ws, out, N, K, S, st), a4 = run<4, false>(Ws, out, N, K, S, st),
              a8 = run<8, false>(Ws, out, N, K, S, st);
        float b1 = run<1, true>(Ws, out, N, K, S, st), b2 = run<2, true>(Ws, out, N, K, S, st), b4 = run<4, true>(Ws, out, N, K, S, st),
              b8 = run<8, true>(Ws, out, N, K, S, st);
        printf("S=%d (%4d WGs) row-major D=1,2,4,8: %6.1f %6.1f %6.1f %6.1f us | tiled: %6.1f %6.1f %6.1f %6.1f us | best %.2f TB/s\n", S, N / 32 * S, a1, a2, a4, a8,
               b1, b2, b4, b8, bytes / 1e6 / (a8 < b8 ? a8 : b8));
    }
    for (int S : {1, 2, 4, 8, 16}) {
        if ((K / S) % 320) continue;
        printf("fragment pattern (16 rows x 64 B per wave load), S=%d (%4d WGs): NG=4,10,20 in flight: %6.1f %6.1f %6.1f us\n", S, N / 64 * S,
               run_frag<4>(Ws, out, N, K, S, st), run_frag<10>(Ws, out, N, K, S, st), run_frag<20>(Ws, out, N, K, S, st));
    }
    {   // operand-fetch phase of the barrier-free skinny GEMM, M = 128 rows of activations (L2-resident)
        const int M = 128;
        float* A;
        hipMalloc(&A, (size_t)M * K * 4);
        hipMemset(A, 0, (size_t)M * K * 4);
        for (int S : {1, 2, 4})
            if ((K / S) % 64 == 0)
                printf("barrier-free GEMM operand fetch, M=128: S=%d (%4d WGs): 2,5,10 groups in flight per wave: %6.1f %6.1f %6.1f us\n", S,
                       (M / 32) * (N / 32) * S, run_gemm_frag<2>(A, Ws, out, M, N, K, S, st), run_gemm_frag<5>(A, Ws, out, M, N, K, S, st),
                       run_gemm_frag<10>(A, Ws, out, M, N, K, S, st));
    }
    return 0;
}
