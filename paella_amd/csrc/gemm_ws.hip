// Weight-streaming fp32 MFMA GEMM for gfx950 -- the kernel the batch-1 / small-batch sampling path lives on.
//
// Regime: C[M,N] = A[M,K] . W[N,K]^T with M = 16..512 activation rows against 6-26 MB of fp32 weights.  At that
// size the GEMM is bound by (a) every CU pulling its share of the weight stream from HBM (a CU sustains only
// ~10 B/clk of HBM: the stream must be spread over all 256 CUs) and (b) per-workgroup latency (a K loop of
// dependent HBM round trips).  Design answers:
//   * waves split N, every wave owns ALL rows of the M tile (BM = 16*TM <= 128): the weight fragment a lane needs
//     for v_mfma_f32_16x16x4_f32 (W[n = lane&15][k0 + 4*(lane>>4) .. +3]) is exactly one 16-byte global load, so
//     W goes HBM -> VGPR directly (no LDS round trip, no barrier on the weight path) through a register ring that
//     keeps 128 k-columns (8 x 16-byte loads per lane) in flight per wave;
//   * only the small, L2-hot activation tile is staged through LDS (XOR-swizzled, conflict-free b128 reads) and
//     shared by the 4 waves; BK grows as BM shrinks so a barrier always covers >= 128 MFMA-k of work;
//   * split-K spreads the weight stream over >= 256-512 workgroups; the reduction is done IN the same launch by
//     the last-arriving workgroup of each output tile (agent-scope release/acquire around a ticket counter,
//     guide section 6 G16), summing the fp32 slabs in fixed slice order -> bit-reproducible, no second launch;
//   * epilogue identical to the tiled kernel (bias, GELU, alpha, residual, timestep scale/shift, remapped stores)
//     plus optional per-16-row column sums of squares (GlobalResponseNorm statistics) so GRN needs no extra pass.
#include "common.h"
#include "gemm_device.h"

template <int TM, int TN, int BK, bool APRO>
__global__ __launch_bounds__(256) void gemm_ws_kernel(GemmArgs g, int kslice, int S, float* __restrict__ slabs,
                                                      unsigned* __restrict__ counters, int tiles_m, int tiles_n) {
    constexpr int BM = TM * 16, BN = 64 * TN;
    constexpr int C4 = BK / 4;                      // 16-byte slots per LDS row
    constexpr int SWZ = (C4 < 16 ? C4 : 16) - 1;    // slot XOR mask
    constexpr int LA = (BM * C4 + 255) / 256;       // A float4 per thread per tile
    constexpr int NG = BK / 16;                     // 16-k groups per tile
    constexpr int PFT = (128 / BK) > 1 ? (128 / BK) : 1;  // W ring depth in tiles (128 k ahead)
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * BK];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid % tiles_m, tile_n = bid / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int slice = blockIdx.y;
    const int kbeg = slice * kslice;
    const int kend = min(g.K, kbeg + kslice);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- weight stream: lane's rows n = n0 + (wave*TN + j)*16 + r16, straight to registers ----
    // Loads are UNCONDITIONAL from clamped in-bounds addresses and their values are never select-masked in the steady
    // state: a load under a lane condition (or a select on its result inside a conditional block) makes hipcc wait for
    // it right away, which serialises the stream.  Out-of-range W rows / A rows only feed output elements that are
    // never stored; the K tail is handled by zeroing the ACTIVATION operand (weights x 0 = 0).
    const float* wrow[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wrow[j] = g.W + (size_t)min(n0 + (wave * TN + j) * 16 + r16, g.N - 1) * g.ldw;
    f32x4 wq[PFT][NG][TN];
    auto load_w = [&](f32x4 (&dst)[NG][TN], int t) {
        const int k0 = kbeg + t * BK + kq * 4;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            const int kc = min(k0 + gg * 16, g.K - 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[gg][j] = *reinterpret_cast<const f32x4*>(wrow[j] + kc);
        }
    };

    // ---- activation tile through LDS ----
    const int arow = tid / C4, ac4 = tid % C4;
    constexpr int RSTEP = 256 / C4;  // rows covered per pass
    f32x4 ra[LA], rs[APRO ? LA : 1], rt;
    rt = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* aptr[LA];
    const float* sptr[APRO ? LA : 1];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int gmc = min(m0 + arow + i * RSTEP, g.M - 1);
        aptr[i] = g.A + (size_t)gmc * g.lda;
        if (APRO) sptr[i] = g.a_scale + (size_t)(gmc / g.a_rows_per_sample) * g.K;
    }
    auto load_a = [&](int t) {
        const int kc = min(kbeg + t * BK + ac4 * 4, g.K - 4);
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + kc);
            if (APRO) rs[i] = *reinterpret_cast<const f32x4*>(sptr[i] + kc);
        }
        if (APRO) rt = *reinterpret_cast<const f32x4*>(g.a_shift + kc);
    };
    auto store_a = [&](int buf, int t) {
        float* As = smem + buf * BM * BK;
        const bool kok = kbeg + t * BK + ac4 * 4 < kend;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int row = arow + i * RSTEP;
            f32x4 v = ra[i];
            if (APRO) v = v * rs[i] + rt;
            if (!kok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (LA * RSTEP == BM || row < BM) *reinterpret_cast<f32x4*>(As + row * BK + ((ac4 ^ (row & SWZ)) << 2)) = v;
        }
    };
    auto compute = [&](int buf, const f32x4 (&wf)[NG][TN]) {
        const float* As = smem + buf * BM * BK;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            f32x4 af[TM];
            const int c4 = gg * 4 + kq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = i * 16 + r16;
                af[i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c4 ^ (row & SWZ)) << 2));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[gg][j][e], af[i][e], acc[i][j], 0, 0, 0);
        }
    };

#pragma unroll
    for (int u = 0; u < PFT; ++u) load_w(wq[u], u);
    load_a(0);
    store_a(0, 0);
    __syncthreads();
    // the loop runs whole chunks of PFT tiles with no per-tile conditionals (one basic block per tile); tiles past
    // the slice end have a zeroed activation operand (store_a masks k >= kend).
    const int nchunks = (ntiles + PFT - 1) / PFT;
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int u = 0; u < PFT; ++u) {
            const int tt = c * PFT + u;
            const int buf = tt & 1;
            load_a(tt + 1);
            __builtin_amdgcn_sched_barrier(0);  // keep the activation prefetch ABOVE the MFMA block (hipcc sinks it otherwise)
            compute(buf, wq[u]);
            load_w(wq[u], tt + PFT);
            __builtin_amdgcn_sched_barrier(0);  // and the weight prefetch above the LDS store / barrier
            store_a(buf ^ 1, tt + 1);
            __syncthreads();
        }
    }

    // ---- split-K: publish the slab, last arriver of the tile reduces in fixed slice order ----
    if (S > 1) {
        constexpr int SLAB = TM * TN * 64 * 4 * 4;  // floats per (tile, slice): fragment order [wave][i][j][lane][4]
        float* my = slabs + ((size_t)bid * S + slice) * SLAB + (size_t)wave * (TM * TN * 64 * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(my + ((i * TN + j) * 64 + lane) * 4) = acc[i][j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* sflag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sflag[0] = __hip_atomic_fetch_add(counters + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (sflag[0] != (unsigned)(S - 1)) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(counters + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
        __syncthreads();
        const float* base = slabs + (size_t)bid * S * SLAB + (size_t)wave * (TM * TN * 64 * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 v = *reinterpret_cast<const f32x4*>(base + ((i * TN + j) * 64 + lane) * 4);
                for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(base + (size_t)s * SLAB + ((i * TN + j) * 64 + lane) * 4);
                acc[i][j] = v;
            }
    }

    // ---- epilogue: lane holds out[m0 + i*16 + r16][n0 + (wave*TN+j)*16 + kq*4 .. +3] ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 16 + r16;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wave * TN + j) * 16 + kq * 4;
            const bool ok = m < g.M && n < g.N;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                v = epilogue_apply(g.ep, g.N, m, n, acc[i][j]);
                epilogue_write(g.ep, g.C, g.ldc, m, n, v);
            }
            if (g.ep.sumsq_out) {  // kernel-uniform
                f32x4 q = v * v;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    q[0] += __shfl_xor(q[0], o, 64);
                    q[1] += __shfl_xor(q[1], o, 64);
                    q[2] += __shfl_xor(q[2], o, 64);
                    q[3] += __shfl_xor(q[3], o, 64);
                }
                if (r16 == 0 && n < g.N && m0 + i * 16 < g.M)
                    *reinterpret_cast<f32x4*>(g.ep.sumsq_out + (size_t)((m0 >> 4) + i) * g.N + n) = q;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int gemm_tile_counters(unsigned** out);  // gemm.hip
int gemm_max_tiles();

template <int TM, int TN, int BK>
static void launch_ws(const GemmArgs& g, int kslice, int S, float* slabs, unsigned* g_counters, int tiles_m, int tiles_n, hipStream_t st) {
    dim3 grid(tiles_m * tiles_n, S);
    if (g.a_scale)
        hipLaunchKernelGGL((gemm_ws_kernel<TM, TN, BK, true>), grid, dim3(256), 0, st, g, kslice, S, slabs, g_counters, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_ws_kernel<TM, TN, BK, false>), grid, dim3(256), 0, st, g, kslice, S, slabs, g_counters, tiles_m, tiles_n);
}

// tm_code: 0..3 -> TM = 1,2,4,8 ; tn in {1,2}
int launch_gemm_ws(const GemmArgs& g, int tm_code, int tn, int splitk, void* ws, size_t ws_bytes, hipStream_t st) {
    unsigned* g_counters = nullptr;
    { const int rc = gemm_tile_counters(&g_counters); if (rc != PAELLA_OK) return rc; }
    const int kMaxTiles = gemm_max_tiles();
    const int TM = 1 << tm_code, BM = 16 * TM, BN = 64 * tn;
    const int BK = TM >= 8 ? 32 : (TM == 4 ? 64 : 128);
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int S = splitk < 1 ? 1 : splitk;
    int kslice = ((g.K + S - 1) / S + BK - 1) / BK * BK;
    S = (g.K + kslice - 1) / kslice;
    if (S > 1) {
        const size_t need = (size_t)tiles_m * tiles_n * S * BM * BN * sizeof(float);
        if (!ws || need > ws_bytes || tiles_m * tiles_n > kMaxTiles) {
            paella_set_error("gemm_ws: split-K workspace too small (%zu needed)", need);
            return PAELLA_ERR_WORKSPACE;
        }
    }
    float* slabs = reinterpret_cast<float*>(ws);
#define WS_CASE(TMv, TNv, BKv) launch_ws<TMv, TNv, BKv>(g, kslice, S, slabs, g_counters, tiles_m, tiles_n, st)
    if (tn == 1) {
        switch (tm_code) {
            case 0: WS_CASE(1, 1, 128); break;
            case 1: WS_CASE(2, 1, 128); break;
            case 2: WS_CASE(4, 1, 64); break;
            default: WS_CASE(8, 1, 32); break;
        }
    } else {
        switch (tm_code) {
            case 0: WS_CASE(1, 2, 128); break;
            case 1: WS_CASE(2, 2, 128); break;
            case 2: WS_CASE(4, 2, 64); break;
            default: WS_CASE(8, 2, 32); break;
        }
    }
#undef WS_CASE
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
