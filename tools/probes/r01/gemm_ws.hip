// Weight-streaming fp32 MFMA GEMM for gfx950: the M <= 64 contractions of the batch-1 sampling path
// (deepest UNet level: 16 positions per sample, cond + uncond rows -> M = 32 against 6-26 MB of fp32 weights).
//
// Regime: C[M,N] = A[M,K] . W[N,K]^T is bound by pulling W from HBM once; the LDS-tiled kernel (gemm.hip) spends as
// many LDS-pipe cycles staging 32x32 tiles as the matrix cores spend multiplying them and reaches ~3.7 TB/s of weight
// stream, a plain streaming read of the same bytes reaches 5.5 TB/s (tools/probes/stream_probe.hip).  Design answers:
//   * waves split N, every wave owns ALL rows of the M tile: the weight fragment a lane needs for
//     v_mfma_f32_16x16x4_f32 (W[n = lane&15][k0 + 4*(lane>>4) .. +3]) is exactly one 16-byte global load, so W goes
//     HBM -> VGPR directly: no LDS round trip, no barrier, and the wave's WHOLE K slice (NKG x 16 columns) is requested
//     up front -- the kernel has no K loop, just "everything in flight, then multiply as it lands" (counted vmcnt);
//   * the activation slice [BM rows x kslice] is small and L2-hot: it is staged ONCE into LDS (XOR-swizzled,
//     conflict-free ds_read_b128 fragments) with the A prologue (GRN scale/shift or LayerNorm-on-load) applied on the way;
//   * split-K spreads the stream over >= 300 workgroups; slabs are tiny at this M (S*M*N*4 bytes) and are combined in the
//     same launch by the last-arriving workgroup in fixed slice order (write-through sc1 stores + relaxed ticket);
//   * epilogue shared with the tiled kernel (bias, GELU, alpha, residual, timestep scale/shift, remapped stores, GRN and
//     LayerNorm statistics for the consumer).
#include "common.h"
#include "gemm_device.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int TM, int NKG, int APRO>  // BM = 16*TM rows, kslice <= 16*NKG columns; APRO: 0 none, 1 GRN scale/shift, 2 LayerNorm from row statistics
__global__ __launch_bounds__(256) void gemm_ws_kernel(GemmArgs g, int kslice, int S, float* __restrict__ slabs,
                                                      unsigned* __restrict__ counters, int tiles_m, int tiles_n, unsigned slab_bytes) {
    constexpr int BM = TM * 16, BN = 64;
    constexpr int KS = NKG * 16;           // floats per LDS row (multiple of 64: the slot XOR stays inside the row)
    constexpr int C4 = KS / 4;             // 16-byte slots per row
    constexpr int LA = BM * C4 / 256;      // float4 per thread to stage the activation slice
    static_assert(NKG % 4 == 0 && (BM * C4) % 256 == 0, "slice geometry");
    __shared__ __attribute__((aligned(16))) float As[BM * KS];
    __shared__ float ln_mu_s[APRO == 2 ? BM : 1], ln_rs_s[APRO == 2 ? BM : 1];

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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    // ---- activation slice: global loads first (L2 hits, and in-order return lets them land ahead of the weight stream) ----
    // Loads are unconditional from clamped in-bounds addresses (a load under a lane condition makes hipcc wait for it at
    // once); rows past M only feed outputs that are never stored, columns past the slice end are zeroed at the LDS store.
    f32x4 ra[LA], rs[APRO == 1 ? LA : 1], rt[APRO == 1 ? LA : 1];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / C4, c4 = idx % C4;
        const int gm = min(m0 + row, g.M - 1);
        const int kc = min(kbeg + c4 * 4, g.K - 4);
        ra[i] = *reinterpret_cast<const f32x4*>(g.A + (size_t)gm * g.lda + kc);
        if (APRO == 1) {
            rs[i] = *reinterpret_cast<const f32x4*>(g.a_scale + (size_t)(gm / g.a_rows_per_sample) * g.K + kc);
            rt[i] = *reinterpret_cast<const f32x4*>(g.a_shift + kc);
        }
    }
    // ---- weight stream: the wave's whole slice, straight to registers ----
    const float* wrow = g.W + (size_t)min(n0 + wave * 16 + r16, g.N - 1) * g.ldw;
    f32x4 wf[NKG];
#pragma unroll
    for (int gg = 0; gg < NKG; ++gg) wf[gg] = *reinterpret_cast<const f32x4*>(wrow + min(kbeg + gg * 16 + kq * 4, kend - 4));

    if (APRO == 2) {
        // LayerNorm-on-load: combine the producer's per-16-column (sum, sumsq) partials of each row (8 lanes per row).
        const int row = tid >> 3, part = tid & 7;
        if (row < BM) {
            const float* stp = g.ln_stats + (size_t)min(m0 + row, g.M - 1) * g.ln_nblk * 2;
            double s = 0.0, q = 0.0;
            for (int j = part; j < g.ln_nblk; j += 8) { s += (double)stp[2 * j]; q += (double)stp[2 * j + 1]; }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            const double mean = s / (double)g.K;
            const double var = q / (double)g.K - mean * mean;
            if (part == 0) {
                ln_mu_s[row] = (float)mean;
                ln_rs_s[row] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)g.ln_eps));
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / C4, c4 = idx % C4;
        f32x4 v = ra[i];
        if (APRO == 1) v = v * rs[i] + rt[i];
        if (APRO == 2) v = (v - ln_mu_s[row]) * ln_rs_s[row];
        if (kbeg + c4 * 4 >= kend) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(As + row * KS + ((c4 ^ (row & 15)) << 2)) = v;
    }
    __syncthreads();

    f32x4 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gg = 0; gg < NKG; ++gg) {
        f32x4 af[TM];
        const int c4 = gg * 4 + kq;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = i * 16 + r16;
            af[i] = *reinterpret_cast<const f32x4*>(As + row * KS + ((c4 ^ (row & 15)) << 2));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[gg][e], af[i][e], acc[i], 0, 0, 0);
    }

    // ---- split-K: write-through (sc1) slab stores, relaxed ticket, last arriver sums in fixed slice order ----
    if (S > 1) {
        constexpr int FR = TM * 64 * 4;  // floats per wave, fragment order [i][lane][4]
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, (int)slab_bytes, 0x00020000);
        const unsigned mybase = (unsigned)((((size_t)bid * S + slice) * (4 * FR) + (size_t)wave * FR) * sizeof(float));
#pragma unroll
        for (int i = 0; i < TM; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i]), rsrc, mybase + (i * 64 + lane) * 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* sflag = reinterpret_cast<unsigned*>(As);
        if (tid == 0) sflag[0] = __hip_atomic_fetch_add(counters + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (sflag[0] != (unsigned)(S - 1)) return;
        if (tid == 0) __hip_atomic_store(counters + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
        const unsigned tbase = (unsigned)(((size_t)bid * S * (4 * FR) + (size_t)wave * FR) * sizeof(float));
        const unsigned sstride = (unsigned)(4 * FR * sizeof(float));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned off = tbase + (i * 64 + lane) * 16;
            f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));
            int s = 1;
            for (; s + 3 < S; s += 4) {  // 4 loads in flight, added in slice order
                const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (s + 0) * sstride, 0, 16));
                const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (s + 1) * sstride, 0, 16));
                const f32x4 a2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (s + 2) * sstride, 0, 16));
                const f32x4 a3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (s + 3) * sstride, 0, 16));
                v += a0; v += a1; v += a2; v += a3;
            }
            for (; s < S; ++s) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + s * sstride, 0, 16));
            acc[i] = v;
        }
    }

    // ---- epilogue: lane holds out[m0 + i*16 + r16][n0 + wave*16 + kq*4 .. +3] ----
    const int nb = n0 + wave * 16;
    const int n = nb + kq * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 16 + r16;
        const bool ok = m < g.M && n < g.N;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            v = epilogue_apply(g.ep, g.N, m, n, acc[i]);
            epilogue_write(g.ep, g.C, g.ldc, m, n, v);
        }
        if (g.ep.sumsq_out) {  // kernel-uniform: per-16-row column sums of squares (GlobalResponseNorm statistics)
            f32x4 q = v * v;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                q[0] += __shfl_xor(q[0], o, 64);
                q[1] += __shfl_xor(q[1], o, 64);
                q[2] += __shfl_xor(q[2], o, 64);
                q[3] += __shfl_xor(q[3], o, 64);
            }
            if (r16 == 0 && n < g.N && m0 + i * 16 < g.M) *reinterpret_cast<f32x4*>(g.ep.sumsq_out + (size_t)((m0 >> 4) + i) * g.N + n) = q;
        }
        if (g.ep.rowstat_out) {  // kernel-uniform: per-row (sum, sum of squares) over this 16-column block (LayerNorm-on-load)
            float rsum = (v[0] + v[1]) + (v[2] + v[3]);
            float rsq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            rsum += __shfl_xor(rsum, 16, 64); rsq += __shfl_xor(rsq, 16, 64);
            rsum += __shfl_xor(rsum, 32, 64); rsq += __shfl_xor(rsq, 32, 64);
            if (kq == 0 && m < g.M && nb < g.N) {
                float* dstp = g.ep.rowstat_out + ((size_t)m * (g.N >> 4) + (nb >> 4)) * 2;
                dstp[0] = rsum; dstp[1] = rsq;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int gemm_tile_counters(unsigned** out, hipStream_t st);  // gemm.hip
int gemm_max_tiles();

template <int TM, int NKG>
static void launch_ws(const GemmArgs& g, int kslice, int S, float* slabs, unsigned* ctr, int tiles_m, int tiles_n, hipStream_t st) {
    const unsigned slab_bytes = S > 1 ? (unsigned)((size_t)tiles_m * tiles_n * S * TM * 1024 * sizeof(float)) : 0u;
    dim3 grid(tiles_m * tiles_n, S);
    if (g.a_scale)
        hipLaunchKernelGGL((gemm_ws_kernel<TM, NKG, 1>), grid, dim3(256), 0, st, g, kslice, S, slabs, ctr, tiles_m, tiles_n, slab_bytes);
    else if (g.ln_stats)
        hipLaunchKernelGGL((gemm_ws_kernel<TM, NKG, 2>), grid, dim3(256), 0, st, g, kslice, S, slabs, ctr, tiles_m, tiles_n, slab_bytes);
    else
        hipLaunchKernelGGL((gemm_ws_kernel<TM, NKG, 0>), grid, dim3(256), 0, st, g, kslice, S, slabs, ctr, tiles_m, tiles_n, slab_bytes);
}

// tm_code 0..2 -> BM = 16, 32, 64; nk_code 0..2 -> K slice capacity 128, 192, 320 columns.  splitk is a lower bound:
// the slice count is raised until a slice fits the kernel's register-resident capacity.
int launch_gemm_ws(const GemmArgs& g, int tm_code, int nk_code, int splitk, void* ws, size_t ws_bytes, hipStream_t st) {
    if (tm_code < 0 || tm_code > 2 || nk_code < 0 || nk_code > 2) { paella_set_error("gemm_ws: bad tile code"); return PAELLA_ERR_ARG; }
    unsigned* ctr = nullptr;
    { const int rc = gemm_tile_counters(&ctr, st); if (rc != PAELLA_OK) return rc; }
    const int BM = 16 << tm_code, BN = 64;
    const int cap = nk_code == 0 ? 128 : (nk_code == 1 ? 192 : 320);
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int S = splitk < 1 ? 1 : splitk;
    if ((g.K + S - 1) / S > cap) S = (g.K + cap - 1) / cap;
    int kslice = ((g.K + S - 1) / S + 15) / 16 * 16;
    S = (g.K + kslice - 1) / kslice;
    if (S > 1) {
        const size_t need = (size_t)tiles_m * tiles_n * S * BM * BN * sizeof(float);
        if (!ws || need > ws_bytes || tiles_m * tiles_n > gemm_max_tiles() || need >= ((size_t)1 << 31)) {
            paella_set_error("gemm_ws: split-K workspace too small (%zu needed)", need);
            return PAELLA_ERR_WORKSPACE;
        }
    }
    float* slabs = reinterpret_cast<float*>(ws);
#define WS_CASE(TMv, NKGv) launch_ws<TMv, NKGv>(g, kslice, S, slabs, ctr, tiles_m, tiles_n, st)
    switch (tm_code * 3 + nk_code) {
        case 0: WS_CASE(1, 8); break;
        case 1: WS_CASE(1, 12); break;
        case 2: WS_CASE(1, 20); break;
        case 3: WS_CASE(2, 8); break;
        case 4: WS_CASE(2, 12); break;
        case 5: WS_CASE(2, 20); break;
        case 6: WS_CASE(4, 8); break;
        case 7: WS_CASE(4, 12); break;
        default: WS_CASE(4, 20); break;
    }
#undef WS_CASE
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
