// OPT-IN fast mode, outside the fp32 parity contract: bf16-operand MFMA GEMM with fp32 accumulation.
//
// C[M,N] = epilogue( prologue(A)[M,K] . W[N,K]^T ) like gemm.hip, but the operands enter the matrix cores as bf16
// (v_mfma_f32_16x16x32_bf16: 16x the fp32 MFMA rate on gfx950), products exact, sums in fp32.  Activations stay fp32 in
// HBM (every other kernel of the path is unchanged); they are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) after the A
// prologue while being staged into LDS.  Weights are converted ONCE into a bf16 shadow copy kept next to the fp32
// original (registry below), which halves the weight stream.  Same tile machinery as the fp32 kernel: 256-thread
// workgroups, BK = 64, register-prefetched double-buffered LDS with the XOR-swizzled 16-byte slots (a slot = 8 bf16),
// XCD-aware tile order, deterministic in-launch split-K combine, shared epilogue.
// Nothing selects this kernel unless the host called paella_set_gemm_precision(1); the argmax-flip rate against the
// fp32 path is reported by tests/test_gpu_fastmode.py and bench.py --gemm bf16.
#include "common.h"
#include <atomic>
#include <mutex>
#include "gemm_device.h"
#include <map>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TN, int APRO>  // APRO: 0 none, 1 GRN scale/shift, 2 LayerNorm from row statistics
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g, const unsigned short* __restrict__ Wb, int kslice, int S,
                                                        float* __restrict__ slabs, int tiles_m, int tiles_n,
                                                        unsigned* __restrict__ counters, unsigned slab_bytes) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 64;
    constexpr int LA = (BM * 8 + 255) / 256, LB = (BN * 8 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int ROWB = BK * 2;  // bytes per LDS row (8 slots of 16 B)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * ROWB];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid % tiles_m, tile_n = bid / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.y * kslice;
    const int kend = min(g.K, kbeg + kslice);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;
    const int ldrow = tid >> 3, ldc = tid & 7;  // staging: 8 threads per row, 8 elements (one 16-byte bf16 slot) each

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Loads are unconditional from clamped in-bounds addresses (see gemm.hip); the K tail is zeroed on the activation side.
    struct Stage { f32x4 a0[LA], a1[LA]; f32x4 s0[APRO == 1 ? LA : 1], s1[APRO == 1 ? LA : 1]; f32x4 t0, t1; u32x4 b[LB]; };
    Stage R[2];
    const float* aptr[LA];
    const float* sptr[APRO == 1 ? LA : 1];
    const unsigned short* bptr[LB];
    float ln_mu[APRO == 2 ? LA : 1], ln_rs[APRO == 2 ? LA : 1];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int gmc = min(m0 + ldrow + i * 32, g.M - 1);
        aptr[i] = g.A + (size_t)gmc * g.lda;
        if (APRO == 1) sptr[i] = g.a_scale + (size_t)(gmc / g.a_rows_per_sample) * g.K;
        if (APRO == 2) {
            const float* stp = g.ln_stats + (size_t)gmc * g.ln_nblk * 2;
            RowStatAcc acc;  // (sum, centred M2) partials per 16-column block: gemm_device.h
            for (int j = ldc; j < g.ln_nblk; j += 8) acc.add(stp[2 * j], stp[2 * j + 1]);
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { acc.S += __shfl_xor(acc.S, o, 64); acc.Q += __shfl_xor(acc.Q, o, 64); acc.M += __shfl_xor(acc.M, o, 64); }
            acc.finish(g.K, g.ln_eps, ln_mu[i], ln_rs[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) bptr[i] = Wb + (size_t)min(n0 + ldrow + i * 32, g.N - 1) * g.ldw;

    auto load_tile = [&](Stage& r, int t) {
        const int kc = min(kbeg + t * BK + ldc * 8, g.K - 8);
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            r.a0[i] = *reinterpret_cast<const f32x4*>(aptr[i] + kc);
            r.a1[i] = *reinterpret_cast<const f32x4*>(aptr[i] + kc + 4);
            if (APRO == 1) {
                r.s0[i] = *reinterpret_cast<const f32x4*>(sptr[i] + kc);
                r.s1[i] = *reinterpret_cast<const f32x4*>(sptr[i] + kc + 4);
            }
        }
        if (APRO == 1) {
            r.t0 = *reinterpret_cast<const f32x4*>(g.a_shift + kc);
            r.t1 = *reinterpret_cast<const f32x4*>(g.a_shift + kc + 4);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) r.b[i] = *reinterpret_cast<const u32x4*>(bptr[i] + kc);
    };
    auto store_tile = [&](const Stage& r, int t) {
        unsigned char* As = smem + (t & 1) * (BM + BN) * ROWB;
        unsigned char* Bs = As + BM * ROWB;
        const bool kok = kbeg + t * BK + ldc * 8 < kend;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int row = ldrow + i * 32;
            f32x4 v0 = r.a0[i], v1 = r.a1[i];
            if (APRO == 1) { v0 = v0 * r.s0[i] + r.t0; v1 = v1 * r.s1[i] + r.t1; }
            if (APRO == 2) { v0 = (v0 - ln_mu[i]) * ln_rs[i]; v1 = (v1 - ln_mu[i]) * ln_rs[i]; }
            if (!kok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
            const f32x8 v = f32x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const bf16x8 h = __builtin_convertvector(v, bf16x8);  // RNE, v_cvt_pk_bf16_f32
            if (LA * 32 == BM || row < BM) *reinterpret_cast<bf16x8*>(As + row * ROWB + ((ldc ^ (row & 7)) << 4)) = h;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int row = ldrow + i * 32;
            if (LB * 32 == BN || row < BN) *reinterpret_cast<u32x4*>(Bs + row * ROWB + ((ldc ^ (row & 7)) << 4)) = r.b[i];
        }
    };
    auto compute = [&](int t) {
        const unsigned char* As = smem + (t & 1) * (BM + BN) * ROWB;
        const unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[TM], bf[TN];
            const int slot = kk * 4 + kq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 16 + r16;
                af[i] = *reinterpret_cast<const bf16x8*>(As + row * ROWB + ((slot ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wn * TN + j) * 16 + r16;
                bf[j] = *reinterpret_cast<const bf16x8*>(Bs + row * ROWB + ((slot ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    const int ntiles = (kend - kbeg + BK - 1) / BK;
    load_tile(R[0], 0);
    load_tile(R[1], 1);
    store_tile(R[0], 0);
    __syncthreads();
    int t = 0;
    for (; t + 2 <= ntiles; t += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            load_tile(R[u], t + u + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute(t + u);
            __builtin_amdgcn_sched_barrier(0);
            store_tile(R[(u + 1) & 1], t + u + 1);
            __syncthreads();
        }
    }
    if (t < ntiles) {  // one tile left; it is in LDS
        compute(t);
        __syncthreads();
    }

    // ---- split-K: write-through slabs, relaxed ticket, last arriver sums in fixed slice order (as gemm.hip) ----
    if (S > 1) {
        constexpr int FR = TM * TN * 64 * 4;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, (int)slab_bytes, 0x00020000);
        const unsigned mybase = (unsigned)((((size_t)bid * S + blockIdx.y) * (4 * FR) + (size_t)wave * FR) * sizeof(float));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsrc, mybase + ((i * TN + j) * 64 + lane) * 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* sflag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) sflag[0] = __hip_atomic_fetch_add(counters + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (sflag[0] != (unsigned)(S - 1)) return;
        if (tid == 0) __hip_atomic_store(counters + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned tbase = (unsigned)(((size_t)bid * S * (4 * FR) + (size_t)wave * FR) * sizeof(float));
        const unsigned sstride = (unsigned)(4 * FR * sizeof(float));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned off = tbase + ((i * TN + j) * 64 + lane) * 16;
                f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16));
                for (int s = 1; s < S; ++s) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + s * sstride, 0, 16));
                acc[i][j] = v;
            }
    }

    // ---- epilogue: lane holds out[m = ..+r16][n = ..+kq*4 .. +3] ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 16 + r16;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + kq * 4;
            const bool ok = m < g.M && n < g.N;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                v = epilogue_apply(g.ep, g.N, m, n, acc[i][j]);
                epilogue_write(g.ep, g.C, g.ldc, m, n, v);
            }
            if (g.ep.sumsq_out) {
                f32x4 q = v * v;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    q[0] += __shfl_xor(q[0], o, 64);
                    q[1] += __shfl_xor(q[1], o, 64);
                    q[2] += __shfl_xor(q[2], o, 64);
                    q[3] += __shfl_xor(q[3], o, 64);
                }
                const int mg = m0 + (wm * TM + i) * 16;
                if (r16 == 0 && n < g.N && mg < g.M) *reinterpret_cast<f32x4*>(g.ep.sumsq_out + (size_t)(mg >> 4) * g.N + n) = q;
            }
            if (g.ep.rowstat_out) {
                float rs, rq;
                rowstat_block(v, rs, rq);
                const int nb = n0 + (wn * TN + j) * 16;
                if (kq == 0 && m < g.M && nb < g.N) {
                    float* dstp = g.ep.rowstat_out + ((size_t)m * (g.N >> 4) + (nb >> 4)) * 2;
                    dstp[0] = rs; dstp[1] = rq;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// bf16 shadow copies of the fp32 weights
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const __bf16 h = (__bf16)src[i];
        dst[i] = __builtin_bit_cast(unsigned short, h);
    }
}

struct ShadowEntry { size_t numel; unsigned short* b; bool dirty; };
static std::map<const float*, ShadowEntry> g_shadow;  // keyed by the fp32 tensor's base address
static std::mutex g_shadow_mu;                       // models on different devices / host threads register and look up concurrently
static std::atomic<int> g_precision{0};                            // 0 = exact fp32 (default), 1 = bf16 operands

// `st` = the stream the fp32 tensor was (re)written on; the copy is complete when this returns
static int shadow_convert(const float* base, ShadowEntry& e, hipStream_t st) {
    if (!e.b) HIP_CHECK_RET(hipMalloc((void**)&e.b, e.numel * sizeof(unsigned short)));
    size_t blocks = (e.numel + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, base, e.b, e.numel);
    LAUNCH_CHECK_RET();
    HIP_CHECK_RET(hipStreamSynchronize(st));
    e.dirty = false;
    return PAELLA_OK;
}

// Models call this for every library-owned weight tensor at finalize (and again after a reload: the copy is refreshed).
int gemm_register_weight(const float* base, size_t numel, hipStream_t st) {
    if (!base || numel == 0) return PAELLA_OK;
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    ShadowEntry& e = g_shadow[base];
    if (e.b && e.numel != numel) { (void)hipFree(e.b); e.b = nullptr; }
    e.numel = numel;
    e.dirty = true;
    return g_precision == 1 ? shadow_convert(base, e, st) : PAELLA_OK;
}

void gemm_unregister_weight(const float* base) {
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    auto it = g_shadow.find(base);
    if (it == g_shadow.end()) return;
    if (it->second.b) (void)hipFree(it->second.b);
    g_shadow.erase(it);
}

int gemm_precision() { return g_precision; }

extern "C" int paella_set_gemm_precision(int mode) {
    if (mode != 0 && mode != 1) { paella_set_error("gemm precision mode must be 0 (fp32) or 1 (bf16 operands)"); return PAELLA_ERR_ARG; }
    if (mode == 1) {
        HIP_CHECK_RET(hipDeviceSynchronize());  // weights may still be in flight on any stream
        std::lock_guard<std::mutex> lock(g_shadow_mu);
        for (auto& kv : g_shadow)
            if (kv.second.dirty || !kv.second.b) { const int rc = shadow_convert(kv.first, kv.second, 0); if (rc != PAELLA_OK) return rc; }
    }
    g_precision = mode;
    return PAELLA_OK;
}

extern "C" int paella_get_gemm_precision(void) { return g_precision; }

// bf16 view of an fp32 weight pointer (any offset into a registered tensor), or null
static const unsigned short* shadow_lookup(const float* W) {
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    auto it = g_shadow.upper_bound(W);
    if (it == g_shadow.begin()) return nullptr;
    --it;
    const float* base = it->first;
    if (W < base || W >= base + it->second.numel || !it->second.b || it->second.dirty) return nullptr;
    return it->second.b + (W - base);
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
static void launch_b(const GemmArgs& g, const unsigned short* Wb, int kslice, int S, float* slabs, unsigned* tickets, hipStream_t st) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, S);
    unsigned* counters = S > 1 ? tickets : nullptr;
    const size_t slab_bytes = (size_t)tiles_m * tiles_n * S * BM * BN * sizeof(float);
    if (g.a_scale)
        hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN, 1>), grid, dim3(256), 0, st, g, Wb, kslice, S, slabs, tiles_m, tiles_n, counters, (unsigned)slab_bytes);
    else if (g.ln_stats)
        hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN, 2>), grid, dim3(256), 0, st, g, Wb, kslice, S, slabs, tiles_m, tiles_n, counters, (unsigned)slab_bytes);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN, 0>), grid, dim3(256), 0, st, g, Wb, kslice, S, slabs, tiles_m, tiles_n, counters, (unsigned)slab_bytes);
}

// tile: 0 = 128x128, 1 = 64x64, 2 = 32x32, -1 = choose.  Returns PAELLA_ERR_STATE when this GEMM cannot take the bf16 path
// (no shadow copy, K % 8 != 0): the caller then runs the fp32 kernel.
int launch_gemm_bf16(const GemmArgs& g, int tile, int splitk, void* ws, size_t ws_bytes, hipStream_t st) {
    const unsigned short* Wb = shadow_lookup(g.W);
    if (!Wb || (g.K & 7) || (g.ldw & 7) || (g.lda & 3) || (g.N & 3) || g.grn_gx || g.ep.grn_gx_out) return PAELLA_ERR_STATE;
    if (g.M <= 0 || g.N <= 0) return PAELLA_OK;
    auto tiles_of = [&](int b) { return (long)((g.M + b - 1) / b) * ((g.N + b - 1) / b); };
    const int ktiles = (g.K + 63) / 64;
    int S = splitk < 1 ? 1 : splitk;
    if (tile < 0) {
        S = 1;
        if (tiles_of(128) >= 512) tile = 0;
        else if (tiles_of(64) >= 512) tile = 1;
        else {
            tile = 2;
            const long t = tiles_of(32);
            while (t * S < 1024 && S < 16 && ktiles / (S * 2) >= 2) S *= 2;
        }
    }
    const int bsz = tile == 0 ? 128 : (tile == 1 ? 64 : 32);
    int kslice = ((g.K + S - 1) / S + 63) / 64 * 64;
    S = (g.K + kslice - 1) / kslice;
    // ws = ticket header (kGemmTicketBytes, see common.h) + slab space
    const bool have_ws = ws && ws_bytes > kGemmTicketBytes;
    const size_t slab_cap = have_ws ? ws_bytes - kGemmTicketBytes : 0;
    while (S > 1 && (!have_ws || (size_t)S * tiles_of(bsz) * bsz * bsz * sizeof(float) > slab_cap || tiles_of(bsz) > (long)kGemmMaxTickets ||
                     (size_t)S * tiles_of(bsz) * bsz * bsz * sizeof(float) >= ((size_t)1 << 31))) {
        S /= 2;
        kslice = ((g.K + S - 1) / S + 63) / 64 * 64;
        S = (g.K + kslice - 1) / kslice;
    }
    unsigned* tickets = have_ws ? reinterpret_cast<unsigned*>(ws) : nullptr;
    float* slabs = have_ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + kGemmTicketBytes) : nullptr;
    switch (tile) {
        case 0: launch_b<2, 2, 4, 4>(g, Wb, kslice, S, slabs, tickets, st); break;
        case 1: launch_b<2, 2, 2, 2>(g, Wb, kslice, S, slabs, tickets, st); break;
        default: launch_b<2, 2, 1, 1>(g, Wb, kslice, S, slabs, tickets, st); break;
    }
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
