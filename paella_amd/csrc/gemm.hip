// fp32 "NT" GEMM for gfx950: C[M,N] = epilogue( prologue(A)[M,K] . W[N,K]^T ).
//
// This one kernel family carries every dense contraction of the Paella hot path
// (reference src/modules.py: nn.Linear :49-53, 1x1 / k2s2 Conv2d :132,155, ConvTranspose2d :174,
// MultiheadAttention in/out projections :10, clf/out_mapper :181,186; src/vqgan.py :17-21,56,66,75,81-87).
//
// Design (MI355X):
//  * exact-fp32 matrix cores: v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain; 157 TF peak).
//  * operands are swapped (MFMA "A" = weight rows, "B" = activation rows) so each lane ends up with
//    4 consecutive output channels -> 16-byte epilogue loads/stores.
//  * 4- or 8-wave workgroups, BK = 32, register-prefetched double-buffered LDS, one barrier per K step; LDS tiles are
//    [rows][32] floats with the 16-byte column slot XOR-swizzled by (row & 7): the staging ds_write_b128 and the
//    fragment ds_read_b128 are conflict-free.
//  * ONE work decomposition for every shape ("stream-K"): the launch's work is the sequence of U = tiles x K-steps
//    units (tile-major, K-step minor); workgroup g of G owns a contiguous range of floor(U/G) or ceil(U/G) units and walks it
//    as ONE prefetch stream, flushing its accumulators whenever the range leaves a tile.
//      - G = tiles           : classic data-parallel (every workgroup one whole tile), large problems;
//      - G = tiles * S       : classic split-K;
//      - any other G         : balanced ranges for skinny problems (batch-1 sampling: M = 32..512 rows against
//                              1280..5120-wide weights) with tiles that cover all of M, so a weight element is fetched
//                              from HBM exactly once and the chip is filled by the K split alone.
//    A tile whose K range is shared by several workgroups is combined deterministically: every part writes an fp32
//    slab in fragment order (write-through sc1 stores, guide G16 R1), takes a relaxed agent-scope ticket, and the last
//    arriver sums the parts in FIXED part order and runs the epilogue -- no float atomics, run-to-run bit-reproducible
//    (the argmax-parity contract needs that).
//  * tickets live in the caller's workspace header (zeroed once by paella_workspace_init, re-armed by the last
//    arriver), so they are per workspace = per model / per device / per captured graph.
//  * XCD-aware workgroup order: workgroup b runs on XCD b%8; logical ids are remapped so that neighbours in unit space
//    (same tile / same weight panel) share an XCD's L2.
#include "common.h"
#include "gemm_device.h"
#include "philox.h"
#include <stdio.h>
#include <stdlib.h>
#include "test_hooks.h"
#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an "unrolled loop" that does not depend on the unroller's size budget
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct SkPlan {
    int tiles_m, tiles_n;  // tile grid
    int KT;                // K steps per tile
    unsigned U;            // tiles_m * tiles_n * KT work units
    unsigned G;            // workgroups of the launch (== gridDim.x; carried here so that it arrives with the rest of the plan in ONE kernel-argument fetch)
    FastDiv dKT, dTM, dQ, dQ1;  // divisions by KT, tiles_m, q, q + 1
    unsigned q, r;         // U = G*q + r: workgroup g owns [g*q + min(g, r), +q + (g < r)) -- 32-bit arithmetic only on the device
    int stagger;           // 256x128 tile: which waves run their non-MFMA phase late (0 none, 1 waves >= 4, 2 odd waves)
    int gm;                // tile rasterisation: groups of gm tile rows, m fastest inside a group, then n, then the next group.
                           // gm >= tiles_m = plain m-fastest order.  The ~32 tiles that run together on an XCD then form a gm x (32 / gm)
                           // block that shares gm activation panels and 32 / gm weight panels in that XCD's L2 instead of 32 + 1.
};

// linear tile index -> (tile_m, tile_n); GROUPED is a compile-time property of the tile (rows >= 64): the skinny batch-1 kernels keep the
// two-instruction plain form (the runtime-selected form cost them 2 % per image)
template <bool GROUPED>
__device__ __forceinline__ void sk_tile_coords(const SkPlan& p, int tile, int& tile_m, int& tile_n) {
    if (!GROUPED || p.gm >= p.tiles_m) {
        tile_n = (int)fast_div((unsigned)tile, p.dTM);
        tile_m = tile - tile_n * p.tiles_m;
    } else {
        const int width = p.gm * p.tiles_n;
        const int grp = tile / width, rem = tile - grp * width;
        const int first_m = grp * p.gm;
        const int gsz = min(p.tiles_m - first_m, p.gm);
        tile_n = rem / gsz;
        tile_m = first_m + (rem - tile_n * gsz);
    }
}

// first unit of workgroup g / the workgroup that owns unit x (inverse of the above)
__device__ __forceinline__ unsigned sk_start(const SkPlan& p, unsigned g) { return g * p.q + min(g, p.r); }
__device__ __forceinline__ unsigned sk_owner(const SkPlan& p, unsigned x) {
    const unsigned big = p.r * (p.q + 1);  // units covered by the r workgroups that own q + 1 units
    return x < big ? fast_div(x, p.dQ1) : p.r + fast_div(x - big, p.dQ);
}

// buffer_load_dwordx4 ... offen lds: 64 lanes x 16 bytes from (descriptor + per-lane offset + uniform offset) straight into LDS at
// lds_wave_base + lane * 16 (the base travels in M0, so it must be wave-uniform).  Device pass only: the host pass has no LDS address space.
// AUX = the load's cache policy bits (0 default, 2 = nt).
template <int AUX = 0>
__device__ __forceinline__ void dma_b128_to_lds(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, AUX);
#endif
}
// Cache policy of the WEIGHT stream of the 32x32 ring tiles (the batch-1 workhorses: every CU reads its weight slices once per launch): nt.  The MI355X guide's
// nt-weights row reports -18 % issue-to-land for such streams; same-box A/B in the model against a probe build with the default policy (tools/ab_nt_weights.sh,
// profiles/r05_ring_nt_weights_ab.txt): 22.68 -> 22.53 ms per image (fp32, three alternating pairs, every pair in the same direction), 17.73 -> 17.31 in the bf16
// fast mode.  Results are bit-identical (a cache hint).  The larger ring tiles (mid-size launches whose weight panels many tile rows re-read from L2) keep the default.
#ifndef PAELLA_RING_W_AUX
#define PAELLA_RING_W_AUX 2
#endif

// one 16x16x32 bf16 MFMA on two 16-byte fragments (8 bf16 each: k = 8 * kq .. + 7 of the fragment's 32-wide k group), fp32 accumulate
__device__ __forceinline__ f32x4 mma_bf16(const f32x4 w, const f32x4 a, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}

#ifdef PAELLA_GEMM_CLOCK_PROBE
__device__ unsigned long long g_clock_probe[2];  // (shader cycles, 100 MHz wall-clock ticks) of one workgroup of the last launch
extern "C" int paella_probe_gemm_clock(unsigned long long* out2) {
    HIP_CHECK_RET(hipDeviceSynchronize());
    HIP_CHECK_RET(hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_clock_probe), 2 * sizeof(unsigned long long)));
    return PAELLA_OK;
}
#endif

// Workgroups per CU a ring instantiation asks the register allocator for: the register-side wish (5 for the 3-stage 32x32 tile, 4 for the other one- and two-block
// wave tiles, 3 for 64x64, 1 for the 8-wave tile) CAPPED by what its LDS footprint admits on a 160 KiB CU -- asking for more than LDS allows only shrinks the register
// budget for nothing (twelve instantiations used to "fail to meet the occupancy target").  Mirrors the kernel's own LDS arithmetic (smem[] below);
// tests/test_kernel_resources.py holds the compiler's numbers against it.
constexpr int ring_lds_bytes(int WM, int WN, int TM, int TN, int APRO, int RING) {
    const int BM = WM * TM * 16, BN = WN * TN * 16;
    const bool big = WM * WN == 8;
    const int aux = (big && APRO == 1) ? 576 : (APRO == 1 ? 512 : (APRO == 4 ? 320 : 0));
    const int scr = big ? 0 : WM * WN * TN * 16;
    return (RING * ((BM + BN) * 32 + aux) + 16 + scr) * 4;
}
constexpr int ring_wg_per_cu(int WM, int WN, int TM, int TN, int APRO, int RING, bool BF) {
    // (the bf16 64x32 tile with the operand-side LayerNorm guard needs > 128 registers: three workgroups per CU there instead of a spill)
    const int wish = WM * WN == 8 ? 1 : TM * TN == 1 ? (RING == 3 ? 5 : 4) : (TM * TN == 2 ? ((BF && APRO == 2 && TM == 2) ? 3 : 4) : 3);
    const int fit = (160 * 1024) / ring_lds_bytes(WM, WN, TM, TN, APRO, RING);
    return fit < wish ? (fit < 1 ? 1 : fit) : wish;
}

template <int WM, int WN, int TM, int TN, int PD, int APRO, bool TAIL = false, int BK = 32, bool DMA = false, int RING = 0, bool BF = false>  // BK: K step (32 or 64 floats per LDS row); APRO: 0 none, 1 GRN scale/shift, 2 LayerNorm from row statistics; TAIL: fused sampling tail (head GEMM);
// DMA: operands that need no transform (W always, A when APRO == 0) go global -> LDS directly (buffer_load ... lds), no staging registers, no ds_write pass
// APRO 4 (ring tiles only): the GRN apply from the producer's UNFINISHED statistics -- a' = a * (1 + gamma * gx / (mean gx + 1e-6)) + shift, the mean
// derived per workgroup from the producer's per-column-tile partial sums (no finalize launch between the two MLP GEMMs).
// RING > 0 (the batch-1 kernels): BOTH operands always go global -> LDS directly into a ring of RING stages with RING - 1 units in flight per workgroup
// (the prefetch depth costs LDS, not registers), and an A-operand prologue is applied to the MFMA FRAGMENTS after they are read back from LDS (GRN scale /
// shift rows ride along in a 2 KB side stage; LayerNorm mean / rstd live in two registers per fragment row).  One barrier per unit, no ds_write at all.
// The 32x32 tile is the batch-1 workhorse and wants 5 workgroups per CU (1280 resident): ask for <= 96 VGPRs there.  Not for the
// GRN-prologue variant (two more staged operands per unit): forced under 96 registers it spills inside the unit loop (measured
// 33 us instead of 25 for 128x1280x5120), so it runs 4 workgroups per CU and the heuristic gives it at most 1024 workgroups.
// The 8-wave 128x64 tiles fit 128 VGPRs without spilling when asked to (126 / 128): two workgroups per CU instead of one.
// BIG = ring variant on 8 waves with 64x64 wave tiles (tile id 36: 256x128): the throughput-regime kernel.  One workgroup per CU, three 48 KiB LDS
// stages (two K steps in flight), fragments read one 16-wide k group ahead of the MFMAs that consume them -- including across the K-step barrier -- so
// neither a ds_read latency nor a store phase ever sits in front of an idle matrix core; the GRN prologue is applied to the fragments from a side
// stage that holds the scale rows of up to 16 consecutive samples.
// BF (the OPT-IN bf16 fast mode, outside the fp32 parity contract): both operands are bf16 in HBM (GemmArgs::A16 / W16) and enter the matrix cores as bf16
// (v_mfma_f32_16x16x32_bf16, fp32 accumulation).  A K step is the same 128 BYTES per row (64 bf16 instead of 32 floats), so the LDS image, the XOR swizzle,
// the LDS-DMA addressing, the ring, the stream-K decomposition, the slabs and every epilogue are shared with the fp32 instantiation: a lane's ds_read_b128 of
// slot kk * 4 + kq holds k = kk * 32 + kq * 8 .. + 7 -- exactly its operand of ONE 16x16x32 MFMA where the fp32 kernel issues four 16x16x4 ones.  Only the
// all-DMA variants exist (direct-to-LDS twins and ring tiles; prologues 0 and 2 -- the folded LayerNorm's operand-side guard re-reads flagged 16-row blocks from the fp32 tensor, see ln_fix).
__global__ __launch_bounds__(64 * WM * WN, RING > 0 ? ring_wg_per_cu(WM, WN, TM, TN, APRO, RING, BF)
                                           : (TAIL && DMA) ? 4  // fused head + tail on the 64x64 direct-to-LDS tile: four independent workgroups per CU
                                           : (TAIL && WM * WN == 8 && TM * TN == 4) ? 4  // fused head + tail on 128x64 tiles: TWO+ workgroups per CU, one's Philox / log epilogue overlaps another's main loop
                                           : (WM == 2 && WN == 2 && TM * TN == 1 && PD == 2 && BK == 32 && (APRO == 0 || APRO == 3)) ? 5  // (GRN / LayerNorm variants spill under 96 registers)
                                           : (DMA && WM * WN == 4 && TM * TN == 16) ? 2  // 128x128 on 4 waves, direct-to-LDS: two independent workgroups per CU (64 KiB of LDS each)
                                           : ((WM * WN == 8 && WM * TM == 8 && WN * TN == 4 && PD == 2 && BK == 32 && APRO == 0 && !TAIL) ? 4 : 1)) void gemm_nt_kernel(GemmArgs g, SkPlan p, float* __restrict__ slabs,
                                                               unsigned* __restrict__ tickets, unsigned slab_bytes) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int SL = BK / 4;     // 16-byte slots per LDS row; slot s of row r lives at s ^ (r & (SL - 1)): conflict-free ds_write_b128 / ds_read_b128
    constexpr int KG = BK / 16;    // 16-wide k groups per K step (4 MFMA k-instructions each)
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int RP = NT / SL;  // rows staged per pass: SL threads (one float4 each) cover a BK-float row
    constexpr int LA = (BM * SL + NT - 1) / NT, LB = (BN * SL + NT - 1) / NT;
    constexpr int TILE_FLOATS = (BM + BN) * BK;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(PD == 1 || PD == 2, "prefetch ring depth 1 or 2");
    static_assert(BK == 32 || BK == 64, "K step of 32 or 64");
    static_assert(!BF || ((DMA || RING > 0) && (APRO == 0 || APRO == 2) && BK == 32), "bf16 operands: direct-to-LDS / ring variants, no operand transform");
    constexpr int ESZ = BF ? 2 : 4;  // bytes per operand element in HBM and LDS
    // Direct-to-LDS operands: one buffer_load_dwordx4 ... lds per wave and 8 tile rows writes 1 KiB at M0 + lane * 16, i.e. LDS stays
    // lane-linear; the XOR swizzle of the 16-byte slots is applied to the SOURCE address instead (lane l of a row fetches chunk
    // (l % 8) ^ (row % 8)).  Needs K % BK == 0 (no activation-side K-tail mask) -- the host picks the register-staged twin otherwise.
    constexpr bool DMA_W = DMA || RING > 0, DMA_A = (DMA && (APRO == 0 || APRO == 2)) || RING > 0;  // (the LayerNorm is folded into the epilogue: the operand stays raw)
    static_assert(!DMA || (PD == 1 && BK == 32 && (BM * SL) % NT == 0 && (BN * SL) % NT == 0), "DMA variant: 1-deep, K step 32, whole passes");
    // PP (RING == 2, bf16 operands only; tile id 37: 256x256): the "ping-pong" throughput tile of the fast mode -- see the PP block in the unit stream below
    constexpr bool PP = RING == 2;
    static_assert(!PP || (BF && NW == 8 && WM == 2 && WN == 4 && TM == 8 && TN == 4 && (APRO == 0 || APRO == 2)), "ping-pong tile: bf16 operands, 2 x 4 waves of 128x64 wave tiles");
    static_assert(RING == 0 || (RING >= 2 && RING <= 4 && !DMA && PD == 1 && BK == 32 && !TAIL && APRO != 3 && (NW == 4 || (TM == 4 && TN == 4 && RING == 3) || PP) && (BM * SL) % NT == 0 && (BN * SL) % NT == 0),
                  "ring variant: 3 or 4 LDS stages (2 for the ping-pong tile), 4 waves (or 8 waves of 64x64 / 128x64 wave tiles), K step 32, whole passes, no implicit convolution");
    static_assert(APRO != 4 || RING > 0, "the GRN-from-raw-statistics prologue exists on ring tiles only");
    constexpr bool BIG = RING > 0 && NW == 8;
    static_assert(!BIG || APRO != 4, "the 8-wave ring tile has no GRN-from-raw-statistics prologue");
    constexpr bool GRN_SIDE = RING > 0 && (APRO == 1 || APRO == 4);
    // ring stage = the A and W tiles + (GRN prologues) a side stage: 8 copies of shift[k0 .. k0 + 32) | scale rows of 8 consecutive samples (APRO 1), or
    // (APRO 4, exec-masked DMAs that move 128 bytes each) shift[k0 .. +32) | gamma[k0 .. +32) | gx rows of 8 consecutive samples
    // (BIG: shift[k0 .. k0 + 32) once (an exec-masked 128-byte DMA) at float 0 | scale rows of 16 consecutive samples from float 64)
    constexpr int AUX_FLOATS = (BIG && APRO == 1) ? 576 : (RING > 0 && APRO == 1) ? 512 : ((RING > 0 && APRO == 4) ? 320 : 0);
    constexpr bool GRN_FIN = RING > 0 && !BIG;  // tiles whose epilogue can finish GlobalResponseNorm's Gx (batch-1 regime only)
    constexpr int GRN_SCR = GRN_FIN ? WM * WN * TN * 16 : 0;  // epilogue scratch: per-wave column sums of squares when a sample spans several waves' rows
    constexpr int STAGE_FLOATS = TILE_FLOATS + AUX_FLOATS;
    constexpr int FLAG_OFF = RING > 0 ? RING * STAGE_FLOATS : 2 * TILE_FLOATS;  // 16 floats for the ticket broadcast behind the stages
    // one LDS object: two tile stages + 16 floats for the ticket broadcast (the NEXT unit's tile is already staged when a
    // segment is flushed, so the flag cannot live inside the stages)
    constexpr int TAIL_FLOATS = TAIL ? BM * WN * 2 : 0;  // fused tail: per row and wave column, the best (score, label)
    __shared__ __attribute__((aligned(16))) float smem[FLAG_OFF + 16 + TAIL_FLOATS + GRN_SCR];

    // ---- this workgroup's unit range ----
    // (everything up to the first operand fetch is latency on every launch: no branch -- the kernel-argument loads of the whole prologue then issue as one
    // batch instead of one round trip per basic block -- and no division by a runtime value)
    const unsigned G = p.G;
    unsigned gid = blockIdx.x;
    {
        const unsigned q = G >> 3, r = G & 7;
        const unsigned xcd = gid & 7, idx = gid >> 3;
        gid = xcd * q + min(xcd, r) + idx;  // XCD x owns the contiguous logical ids [x*q + min(x, r), ...): q + 1 of them when x < r
    }
    const unsigned u0 = sk_start(p, gid);
    const int n = (int)(p.q + (gid < p.r ? 1u : 0u));
    __builtin_assume(n > 0);  // host keeps G <= U, so every workgroup owns at least one unit
#ifdef PAELLA_GEMM_CLOCK_PROBE  // tools/probes/gemm_clock_probe.py builds its own library with this: shader clock the launch really ran at
    const unsigned long long cp_t0 = __builtin_readcyclecounter(), cp_w0 = wall_clock64();
#endif
    const int KT = p.KT;

    const int tid = threadIdx.x;
    const int lane_k = tid & 63;
    const int lane = lane_k;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> SGPR arithmetic for wm / wn / slab bases
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;
    const int ldrow = tid / SL, ldc4 = tid % SL;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Global -> register ring -> LDS.  The ring holds PD units in flight per workgroup.  Loads are unconditional from
    // clamped in-bounds addresses and never select-masked (a load under a lane condition makes hipcc wait for it at
    // once); out-of-range rows only feed outputs that are never stored, and the K tail is zeroed on the ACTIVATION side
    // only when the tile is written to LDS.  Everything the LDS store needs travels with the stage, because the load
    // cursor runs PD units ahead of the compute cursor and may already be in the next tile.
    struct Stage {
        f32x4 a[LA];
        f32x4 s[APRO == 1 ? LA : 1];
        f32x4 t;
        f32x4 b[LB];
        bool kok;
        bool ok[APRO == 3 ? LA : 1];  // implicit conv: the tap of this K step falls inside the input grid for row i
    };
    Stage R[PD];

    // ---- load cursor ----
    // Operands are read through buffer descriptors: per-thread 32-bit byte offsets (recomputed only when the cursor enters a new
    // tile) + ONE uniform K offset in an SGPR per unit -> no vector address arithmetic in the unit loop.  Reads past the end of a
    // buffer return 0 (hardware range check), so only the M / N clamps remain; the K tail is masked when the tile is staged.
    int ltile = (int)fast_div(u0, p.dKT);
    int lkt = (int)(u0 - (unsigned)ltile * (unsigned)KT);
    auto rsrc_of = [](const void* ptr, size_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
    };
    const char* const Abase = BF ? reinterpret_cast<const char*>(g.A16) : reinterpret_cast<const char*>(g.A);
    const char* const Wbase = BF ? reinterpret_cast<const char*>(g.W16) : reinterpret_cast<const char*>(g.W);
    // descriptors are re-based on every tile (first row of the tile / first image of the tile): per-thread offsets stay far below
    // 4 GiB whatever the operand size, and the range check still ends at the true end of each operand
    const size_t a_bytes = APRO == 3 ? (size_t)(g.M / (g.cv.Ho * g.cv.Wo)) * g.cv.Hi * g.cv.Wi * g.cv.C * sizeof(float)
                                     : ((size_t)(g.M - 1) * g.lda + g.K) * ESZ;
    const size_t w_bytes = ((size_t)(g.N - 1) * g.ldw + g.K) * ESZ;
    __amdgpu_buffer_rsrc_t rsrcA = rsrc_of(Abase, a_bytes), rsrcW = rsrc_of(Wbase, w_bytes);
    __amdgpu_buffer_rsrc_t rsrcS = rsrc_of(g.A, 16);
    const __amdgpu_buffer_rsrc_t rsrcT = rsrc_of((APRO == 1 || APRO == 4) ? g.a_shift : g.A, (APRO == 1 || APRO == 4) ? (size_t)g.K * sizeof(float) : 16);
    const __amdgpu_buffer_rsrc_t rsrcG = rsrc_of(APRO == 4 ? g.grn_gamma : g.A, APRO == 4 ? (size_t)g.K * sizeof(float) : 16);
    unsigned aoff[LA], soff[APRO == 1 ? LA : 1], boff[LB];
    unsigned aux_s_off2 = 0;  // (BIG: the same for samples 8..15 of the tile)
    unsigned aux_s_off = 0;  // ring + GRN prologue: this lane's source offset in the scale rows of the tile's samples (lane -> sample lane / 8, 16-byte chunk lane % 8)
    int cy[APRO == 3 ? LA : 1], cx[APRO == 3 ? LA : 1];  // implicit conv: top-left input coordinate of row i (aoff[i] = image base position)
    int ltap = 0, lc0 = 0;                                  // implicit conv: tap and channel offset of the load cursor's K step
    auto set_tile = [&](int tile) __attribute__((always_inline)) {
        int tile_m, tile_n;
        sk_tile_coords<(BM >= 64)>(p, tile, tile_m, tile_n);
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        size_t a_base;  // bytes from g.A to this tile's descriptor base
        int img0 = 0, smp0 = 0;
        if (APRO == 3) {
            img0 = m0 / (g.cv.Ho * g.cv.Wo);
            a_base = (size_t)img0 * g.cv.Hi * g.cv.Wi * g.cv.C * sizeof(float);
        } else {
            a_base = (size_t)m0 * g.lda * ESZ;
        }
        rsrcA = rsrc_of(Abase + a_base, a_bytes - a_base);
        const size_t w_base = (size_t)n0 * g.ldw * ESZ;
        rsrcW = rsrc_of(Wbase + w_base, w_bytes - w_base);
        if (APRO == 1 || APRO == 4) {
            smp0 = (int)fast_div((unsigned)m0, g.a_rps_div);
            const size_t s_bytes = (size_t)(fast_div((unsigned)(g.M - 1), g.a_rps_div) + 1) * g.K * sizeof(float), s_base = (size_t)smp0 * g.K * sizeof(float);
            rsrcS = rsrc_of((APRO == 4 ? g.grn_gx : g.a_scale) + (size_t)smp0 * g.K, s_bytes - s_base);
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int gmc = min(m0 + ldrow + i * RP, g.M - 1);
            if (APRO == 3) {
                const int hw = g.cv.Ho * g.cv.Wo;
                const int bimg = gmc / hw, rem = gmc - bimg * hw;
                const int yo = rem / g.cv.Wo, xo = rem - yo * g.cv.Wo;
                cy[i] = yo * g.cv.stride;
                cx[i] = xo * g.cv.stride;
                aoff[i] = (unsigned)(bimg - img0) * (unsigned)(g.cv.Hi * g.cv.Wi);  // position index of the image's (0, 0), relative to the tile's first image
            } else {
                aoff[i] = (unsigned)(gmc - m0) * (unsigned)g.lda * (unsigned)ESZ + (unsigned)((DMA_A ? (ldc4 ^ (ldrow & (SL - 1))) : ldc4) * 16);  // (row, 16-byte slot) in bytes
            }
            if (APRO == 1 && RING == 0) soff[i] = ((unsigned)((int)fast_div((unsigned)gmc, g.a_rps_div) - smp0) * (unsigned)g.K + (unsigned)(ldc4 * 4)) * 4u;
        }
        if (GRN_SIDE) {
            const int last = (int)fast_div((unsigned)(g.M - 1), g.a_rps_div) - smp0;  // clamp: rows past the last sample re-read it (never used)
            aux_s_off = ((unsigned)min(lane_k >> 3, last) * (unsigned)g.K + (unsigned)((lane_k & 7) * 4)) * 4u;
            if (BIG) aux_s_off2 = ((unsigned)min(8 + (lane_k >> 3), last) * (unsigned)g.K + (unsigned)((lane_k & 7) * 4)) * 4u;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) boff[i] = (unsigned)(min(n0 + ldrow + i * RP, g.N - 1) - n0) * (unsigned)g.ldw * (unsigned)ESZ + (unsigned)((DMA_W ? (ldc4 ^ (ldrow & (SL - 1))) : ldc4) * 16);
    };
    set_tile(ltile);
    if (APRO == 3) {
        ltap = (lkt * BK) / g.cv.C;
        lc0 = lkt * BK - ltap * g.cv.C;
    }
    // LayerNorm-on-load, FOLDED INTO THE EPILOGUE: with mu / rstd the statistics of output row m,
    //     sum_k ((a[m][k] - mu) * rstd) * W[n][k]  =  rstd * (sum_k a[m][k] * W[n][k]  -  mu * wsum[n]),      wsum[n] = sum_k W[n][k]  (precomputed, GemmArgs::ln_wsum)
    // so the main loop multiplies the RAW operand (same loads, DMA and MFMA stream as a plain GEMM -- the per-unit (a - mu) * rstd on the fragments sat between the
    // fragment reads and the MFMAs of every unit and cost 3-4 us per batch-1 launch, 10 % of the matrix-core rate at large batch) and the tile's finisher applies
    // the two per-row scalars to the accumulators.  fr_mu / fr_rs: statistics of this lane's fragment rows (row r16 of 16-row block i), set behind the first operand
    // fetches by ln_row_stats() below.  Computed ONCE: the host only launches this variant with ranges that never change tile_m.
    float fr_mu[APRO == 2 ? TM : 1], fr_rs[APRO == 2 ? TM : 1];
    // the fold cancels when |mu| >> std (gemm_device.h: kLnFoldMaxRatio): such 16-row blocks normalise their operand FRAGMENTS instead (ln_fix) and skip the fold
    float fr_mu_lo[APRO == 2 ? TM : 1];  // mean - (float)mean: the operand-side form subtracts the mean in two pieces (an fp32 mean alone is off by eps * |mean|, i.e. eps * ratio in units of std)
    bool ln_dir[APRO == 2 ? TM : 1];
    // bf16 operands, two regimes (the LDS image is the ROUNDED copy, so a flagged row needs its fp32 source): with the row pre-pass (GemmArgs::ln_row, >= 2048 rows
    // and every 8-wave tile) launch_ln_rowstat_finalize has ALREADY rewritten the flagged rows of the bf16 copy as bf16(LayerNorm(row)) -- the main loop is untouched
    // and the epilogue skips the fold per ROW (ln_pre, per lane); without it (batch-1 launches on the 4-wave tiles) flagged 16-row blocks re-read the fp32 rows in
    // ln_fix.  The 8-wave bf16 tiles have no in-kernel fix (it spills there): the launcher always gives them the pre-pass.
    constexpr bool BF_FIX = BF && APRO == 2 && NW == 4;
    // (a pre-normalised row carries (mu, rstd) = (0, 1) from ln_row_stats on: rstd * (acc - 0 * wsum) is acc, bit for bit -- no per-lane flag array across the epilogue)
    bool ln_any = false;
    int ln_row0 = 0;  // first row of the (only) tile row this launch's LayerNorm statistics belong to
    const int ln_tile0 = ltile;
    auto ln_row_stats = [&]() __attribute__((always_inline)) {
        if constexpr (APRO == 2) {
            // the 4 lanes that hold one fragment row (kq = 0..3) split the producer's per-16-column (sum, M2) pairs as 16-byte chunks (two blocks each; a last odd
            // block as a pair) and xor-reduce; combined in fp64 (RowStatAcc)
            int ln_tm, ln_tn;
            sk_tile_coords<(BM >= 64)>(p, ln_tile0, ln_tm, ln_tn);
            ln_row0 = ln_tm * BM;
            // (the 8-wave bf16 tiles call this BEHIND their main loop: the lane id is re-derived from the hardware there -- a value carried across a main loop that
            // uses every register is spilled -- and laundered, so that no row address is hoisted above the loop either)
            int lane_l = (NW == 8 && BF) ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : lane_k;
            asm volatile("" : "+v"(lane_l));
            const int r16 = lane_l & 15;
            if (PP || g.ln_row) {  // finished once per row by launch_ln_rowstat_finalize (throughput regime; ALWAYS on the 8-wave bf16 tiles -- the ping-pong tile does not even carry the other path): one 16-byte load per fragment row
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int gmc = min(ln_tm * BM + (wm * TM + i) * 16 + r16, g.M - 1);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(g.ln_row + (size_t)gmc * 4);
                    fr_mu[i] = v[0]; fr_rs[i] = v[1]; fr_mu_lo[i] = v[2];
                    if constexpr (BF) {
                        ln_dir[i] = false;
                        const bool pre = g.A != nullptr && v[3] > g.ln_fold_ratio;  // this lane's row was normalised by the pre-pass: no fold for it
                        if (pre) { fr_mu[i] = 0.f; fr_rs[i] = 1.f; }
                        if (g.ln_guard_count) ln_any = ln_any || __builtin_amdgcn_ballot_w64(pre) != 0;  // (test counter only)
                    } else {
                        ln_dir[i] = __builtin_amdgcn_ballot_w64(v[3] > g.ln_fold_ratio) != 0;
                        ln_any = ln_any || ln_dir[i];
                    }
                }
                if (g.ln_guard_count && ln_any && lane_l == 0) atomicAdd(g.ln_guard_count, 1u);  // test hook only (null in the product)
                return;
            }
            const int nch = g.ln_nblk >> 1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int gmc = min(ln_tm * BM + (wm * TM + i) * 16 + r16, g.M - 1);
                const float* st0 = g.ln_stats + (size_t)gmc * g.ln_nblk * 2;
                const bool vec = (((size_t)gmc * g.ln_nblk) & 1) == 0;  // 16-byte aligned row of pairs (always when ln_nblk is even)
                RowStatAcc acc;
                if (vec) {
                    const f32x4* stp = reinterpret_cast<const f32x4*>(st0);
                    int j = kq;
                    for (; j + 12 < nch; j += 16) {  // 4 independent loads in flight
                        const f32x4 v0 = stp[j], v1 = stp[j + 4], v2 = stp[j + 8], v3 = stp[j + 12];
                        acc.add(v0[0], v0[1]); acc.add(v0[2], v0[3]); acc.add(v1[0], v1[1]); acc.add(v1[2], v1[3]);
                        acc.add(v2[0], v2[1]); acc.add(v2[2], v2[3]); acc.add(v3[0], v3[1]); acc.add(v3[2], v3[3]);
                    }
                    for (; j < nch; j += 4) {
                        const f32x4 v0 = stp[j];
                        acc.add(v0[0], v0[1]); acc.add(v0[2], v0[3]);
                    }
                    if ((g.ln_nblk & 1) && kq == 0) acc.add(st0[2 * (g.ln_nblk - 1)], st0[2 * (g.ln_nblk - 1) + 1]);
                } else {
                    for (int j = kq; j < g.ln_nblk; j += 4) acc.add(st0[2 * j], st0[2 * j + 1]);
                }
                acc.S += __shfl_xor(acc.S, 16, 64); acc.Q += __shfl_xor(acc.Q, 16, 64); acc.M += __shfl_xor(acc.M, 16, 64);
                acc.S += __shfl_xor(acc.S, 32, 64); acc.Q += __shfl_xor(acc.Q, 32, 64); acc.M += __shfl_xor(acc.M, 32, 64);
                acc.finish(g.K, g.ln_eps, fr_mu[i], fr_rs[i]);
                fr_mu_lo[i] = (float)(acc.S / (double)g.K - (double)fr_mu[i]);
                ln_dir[i] = (!BF || (BF_FIX && g.A != nullptr)) && __builtin_amdgcn_ballot_w64(fabsf(fr_mu[i]) * fr_rs[i] > g.ln_fold_ratio) != 0;  // wave-uniform, a function of the block's 16 rows only
                ln_any = ln_any || ln_dir[i];
            }
            if (g.ln_guard_count && ln_any && lane_k == 0) atomicAdd(g.ln_guard_count, 1u);  // test hook only (null in the product)
        }
    };
    // operand-side LayerNorm of an A fragment (row block i, 4 consecutive k from kbase) -- only for blocks flagged by ln_row_stats; zero past K like the staged K tail
    // BF (bf16 operands): the LDS image holds the operand ALREADY ROUNDED to bf16 -- at |mu| / std = r that rounding is r * 2^-9 of a standard deviation per element
    // (2-3 % at r = 16, 20-30 % at r = 160: ADVICE r05), and no arithmetic on the rounded value brings it back.  A flagged block therefore re-reads its rows from
    // the fp32 residual stream (GemmArgs::A, the tensor the bf16 copy was made from), normalises in fp32 and rounds the NORMALISED value to bf16: the error is the
    // ordinary 2^-9 of a unit-scale operand again.  kbase counts fp32 K steps like the callers' LDS slots: the fragment's 8 consecutive k start at 2 * kbase.
    // 16 lanes x 32 B per row and fragment, uncoalesced and straight from L2 / HBM: slow, and only ever executed by flagged blocks.
    auto ln_fix = [&](f32x4& a, int i, int kbase) __attribute__((always_inline)) {
        if constexpr (APRO == 2 && !BF) {
            if (ln_dir[i]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (kbase + e < g.K) ? ((a[e] - fr_mu[i]) - fr_mu_lo[i]) * fr_rs[i] : 0.f;
            }
        }
        if constexpr (BF_FIX) {
            if (ln_dir[i]) {
                const int gm = min(ln_row0 + (wm * TM + i) * 16 + r16, g.M - 1);
                const float* src = g.A + (size_t)gm * g.lda + 2 * kbase;
                typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((v[e] - fr_mu[i]) - fr_mu_lo[i]) * fr_rs[i];
                const bf16x4_t lo = __builtin_convertvector(v, bf16x4_t);
                v = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((v[e] - fr_mu[i]) - fr_mu_lo[i]) * fr_rs[i];
                const bf16x4_t hi = __builtin_convertvector(v, bf16x4_t);
                a = __builtin_bit_cast(f32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
    };
    int kt_cur = 0;  // K step of the unit being multiplied (non-ring loops; the ring / DMA tiles need K % BK == 0 and pass kbase = 0)

    auto load_unit = [&](Stage& r, int dma_slot) __attribute__((always_inline)) {  // loads the unit under the load cursor (DMA operands: into LDS stage dma_slot)
        float* dAs = smem + dma_slot * (RING > 0 ? STAGE_FLOATS : TILE_FLOATS) + (wave * (64 / SL)) * BK;  // this wave's first row group of the stage (wave-uniform -> M0)
        float* dBs = dAs + BM * BK;
        const int kofs = lkt * (BK * 4);  // uniform byte offset of this K step -> the loads' SGPR offset
        if constexpr (RING > 0) {
#pragma unroll
            for (int i = 0; i < LA; ++i) dma_b128_to_lds(rsrcA, dAs + i * RP * BK, aoff[i], kofs);
#pragma unroll
            for (int i = 0; i < LB; ++i) dma_b128_to_lds<((BIG || TM * TN != 1) ? 0 : PAELLA_RING_W_AUX)>(rsrcW, dBs + i * RP * BK, boff[i], kofs);
            if (GRN_SIDE && wave == NW - 1) {  // the side stage: 1 KB of shift (8 copies of the 128-byte row) [, 1 KB of gamma], 1 KB of scale / gx rows
                float* dX = smem + dma_slot * STAGE_FLOATS + TILE_FLOATS;
                if (BIG) {  // shift row by lanes 0..7, then two DMAs of 8 sample rows each
                    if (lane_k < 8) dma_b128_to_lds(rsrcT, dX, (unsigned)(lane_k * 16), kofs);
                    dma_b128_to_lds(rsrcS, dX + 64, aux_s_off, kofs);
                    dma_b128_to_lds(rsrcS, dX + 64 + 256, aux_s_off2, kofs);
                } else if (APRO == 4) {  // lanes 0..7 only: one 128-byte row each of shift and gamma (an LDS-DMA writes M0 + lane * 16 for the ACTIVE lanes)
                    if (lane_k < 8) {
                        dma_b128_to_lds(rsrcT, dX, (unsigned)(lane_k * 16), kofs);
                        dma_b128_to_lds(rsrcG, dX + 32, (unsigned)(lane_k * 16), kofs);
                    }
                    dma_b128_to_lds(rsrcS, dX + 64, aux_s_off, kofs);
                } else {
                    dma_b128_to_lds(rsrcT, dX, (unsigned)((lane_k & 7) * 16), kofs);
                    dma_b128_to_lds(rsrcS, dX + 256, aux_s_off, kofs);
                }
            }
            return;
        }
        int oy = 0, ox = 0;
        if (APRO == 3) {
            oy = g.cv.oy0 + g.cv.tsign * (ltap >> g.cv.tw_log2);  // uniform scalar arithmetic
            ox = g.cv.ox0 + g.cv.tsign * (ltap & ((1 << g.cv.tw_log2) - 1));
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            if (APRO == 3) {
                const int yy = cy[i] + oy, xx = cx[i] + ox;
                r.ok[i] = (unsigned)yy < (unsigned)g.cv.Hi && (unsigned)xx < (unsigned)g.cv.Wi;
                const unsigned pos = aoff[i] + (unsigned)(min(max(yy, 0), g.cv.Hi - 1) * g.cv.Wi + min(max(xx, 0), g.cv.Wi - 1));
                r.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, (pos * (unsigned)g.cv.C + (unsigned)(ldc4 * 4)) * 4u, lc0 * 4, 0));
            } else if (DMA_A) {
                dma_b128_to_lds(rsrcA, dAs + i * RP * BK, aoff[i], kofs);
            } else {
                r.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, aoff[i], kofs, 0));
            }
            if (APRO == 1) r.s[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, soff[i], kofs, 0));
        }
        if (APRO == 1) r.t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcT, (unsigned)(ldc4 * 16), kofs, 0));
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            if (DMA_W) dma_b128_to_lds(rsrcW, dBs + i * RP * BK, boff[i], kofs);
            else r.b[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcW, boff[i], kofs, 0));
        }
        r.kok = lkt * BK + ldc4 * 4 < g.K;
    };
    auto store_unit = [&](const Stage& r, int slot) __attribute__((always_inline)) {
        float* As = smem + slot * TILE_FLOATS;
        float* Bs = As + BM * BK;
#pragma unroll
        for (int i = 0; i < (DMA_A ? 0 : LA); ++i) {
            const int row = ldrow + i * RP;
            f32x4 v = r.a[i];
            if (APRO == 1) v = v * r.s[i] + r.t;
            if (!r.kok || (APRO == 3 && !r.ok[i])) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (LA * RP == BM || row < BM) *reinterpret_cast<f32x4*>(As + row * BK + ((ldc4 ^ (row & (SL - 1))) << 2)) = v;
        }
#pragma unroll
        for (int i = 0; i < (DMA_W ? 0 : LB); ++i) {
            const int row = ldrow + i * RP;
            if (LB * RP == BN || row < BN) *reinterpret_cast<f32x4*>(Bs + row * BK + ((ldc4 ^ (row & (SL - 1))) << 2)) = r.b[i];
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA'd part of the stage has landed before the barrier that publishes it
    };
    // The 8-wave 128x128 tiles run one workgroup per CU, two waves per SIMD in lock step: with "store, barrier, read all fragments,
    // multiply" the LDS store phase and the fragment-read latency idle the matrix cores for ~25 % of the launch
    // (profiles/r02_pmc_mfma_busy_config3.txt).  PIPE splits a unit's MFMA block around the LDS hand-over instead:
    //   read this unit's second-k-group fragments | MFMA k-group 0 | MFMA half of k-group 1 | store the next unit's tile, barrier |
    //   read the NEXT unit's first-k-group fragments | MFMA rest of k-group 1
    // so every ds_read runs under MFMAs that do not need it and the store sits between two MFMA runs; same 48 fragment registers.
    constexpr bool PIPE = (NW == 8 && PD == 1 && TM * TN == 8 && BK == 32 && !TAIL);
    f32x4 Fa[KG][TM], Fb[KG][TN];  // PIPE only: fragments by k group; group 0 belongs to the unit ahead during a unit's last quarter
    auto read_group = [&](auto kk_tag, int slot, int kt) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_tag)::value;
        const float* As = smem + slot * TILE_FLOATS;
        const float* Bs = As + BM * BK;
        const int c4 = kk * 4 + kq;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wm * TM + i) * 16 + r16;
            Fa[kk][i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
        }
        if (APRO == 2 && ln_any) {
#pragma unroll
            for (int i = 0; i < TM; ++i) ln_fix(Fa[kk][i], i, kt * BK + c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = (wn * TN + j) * 16 + r16;
            Fb[kk][j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
        }
    };
    auto mfma_group = [&](auto kk_tag, auto e0_tag, auto e1_tag) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_tag)::value, e0 = decltype(e0_tag)::value, e1 = decltype(e1_tag)::value;
        if constexpr (BF) {  // ONE MFMA per (i, j) and k group: the [e0, e1) quarter range selects that share of the TM * TN accumulators
#pragma unroll
            for (int ij = e0 * TM * TN / 4; ij < e1 * TM * TN / 4; ++ij) acc[ij / TN][ij % TN] = mma_bf16(Fb[kk][ij % TN], Fa[kk][ij / TN], acc[ij / TN][ij % TN]);
        } else {
#pragma unroll
            for (int e = e0; e < e1; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Fb[kk][j][e], Fa[kk][i][e], acc[i][j], 0, 0, 0);
        }
    };
    // A 1x1 wave tile alternates two accumulators so its MFMAs are never back-to-back dependent
    // (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency).
    constexpr bool DUAL = (TM * TN == 1);
    f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int slot) __attribute__((always_inline)) {
        const float* As = smem + slot * TILE_FLOATS;
        const float* Bs = As + BM * BK;
        // small wave tiles and the 8-wave 128x128 configs: all fragment reads of the tile up front (one exposed LDS latency
        // per tile); the other big ones: per 16-k group (VGPRs)
        constexpr bool FRAG_FIRST = (TM * TN <= 2) || (NW == 8 && TM + TN <= 6 && PD == 1);
        if constexpr (FRAG_FIRST) {
            f32x4 af[KG][TM], bf[KG][TN];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                const int c4 = kk * 4 + kq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 16 + r16;
                    af[kk][i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 16 + r16;
                    bf[kk][j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
            }
            if (APRO == 2 && ln_any) {
#pragma unroll
                for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i) ln_fix(af[kk][i], i, kt_cur * BK + (kk * 4 + kq) * 4);
            }
            if constexpr (BF) {
                if (DUAL) {
#pragma unroll
                    for (int kk = 0; kk < KG; kk += 2) {
                        acc[0][0] = mma_bf16(bf[kk][0], af[kk][0], acc[0][0]);
                        acc2 = mma_bf16(bf[kk + 1][0], af[kk + 1][0], acc2);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc[i][j] = mma_bf16(bf[kk][j], af[kk][i], acc[i][j]);
                }
            } else if (DUAL) {
#pragma unroll
                for (int kk = 0; kk < KG; kk += 2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kk][0][e], af[kk][0][e], acc[0][0], 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kk + 1][0][e], af[kk + 1][0][e], acc2, 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kk][j][e], af[kk][i][e], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                f32x4 af[TM], bf[TN];
                const int c4 = kk * 4 + kq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 16 + r16;
                    af[i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 16 + r16;
                    bf[j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
                if (APRO == 2 && ln_any) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) ln_fix(af[i], i, kt_cur * BK + c4 * 4);
                }
                if constexpr (BF) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = mma_bf16(bf[j], af[i], acc[i][j]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    // Fused tail: the noise of a logit, log q = log(-log u) with u from Philox keyed by (seed, GLOBAL row, label quad, step), does not depend on the logit -- it is drawn
    // AHEAD of the main loop, under the tile's first operand fetch (a head launch is one whole tile per workgroup), instead of in the epilogue: twenty 64-bit multiplies per
    // label quad and two hardware logs per label, ~4 ms of VALU work per launch at configs[2] that used to follow the 7.3 ms of MFMAs.  Same counters, same arithmetic
    // (philox.h: one definition for the fused and the unfused tail): identical tokens.
    constexpr bool TAIL_AHEAD = TAIL && DMA;  // the product's head tiles (64x64 direct-to-LDS, fp32 and bf16); the A/B-only head tiles sit at their register limit and keep drawing per fragment in the epilogue
    float tlq[TAIL_AHEAD ? TM * TN : 1][4];
    int tlq_tile = -1;       // tile the drawn noise belongs to
    int tail_cur_tile = -1;  // tile being finished (set by flush)
    auto tail_draw = [&](int tile) __attribute__((always_inline)) {
        if constexpr (TAIL_AHEAD) {
            if (g.ft.mode != 1) {
                const FusedTail& ft = g.ft;
                int tile_m, tile_n;
                sk_tile_coords<(BM >= 64)>(p, tile, tile_m, tile_n);
                const uint64_t seed = ft.seed + (ft.seed_ptr ? *ft.seed_ptr : 0ull);
                const int64_t row_off = ft.row_offset + (ft.row_offset_ptr ? *ft.row_offset_ptr : 0);
                const int L4 = g.N >> 2;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = tile_m * BM + (wm * TM + i) * 16 + r16;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int nn = tile_n * BN + (wn * TN + j) * 16 + kq * 4;
                        uint32_t rb[4];
                        philox4x32(seed, (uint64_t)(m + row_off) * L4 + (nn >> 2), ft.offset, rb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) tlq[i * TN + j][e] = log_exp1(rb[e]);
                    }
                }
                tlq_tile = tile;
            }
        }
    };
    // ---- epilogue of a finished tile: lane holds out[m = ..+r16][n = ..+kq*4 .. +3] ----
    auto epilogue = [&](int m0, int n0, int r16, int kq) {
        if constexpr (TAIL) {
            // ---- fused sampling tail: logits never leave the registers ----
#pragma clang fp contract(off)
            const FusedTail& ft = g.ft;
            const uint64_t seed = ft.seed + (ft.seed_ptr ? *ft.seed_ptr : 0ull);
            const int64_t row_off = ft.row_offset + (ft.row_offset_ptr ? *ft.row_offset_ptr : 0);
            const int L4 = g.N >> 2;
            if (TAIL_AHEAD && tlq_tile != tail_cur_tile) tail_draw(tail_cur_tile);  // (never in the product: a head launch is one tile per workgroup and drew its noise under the first fetch)
            float* s_score = smem + FLAG_OFF + 16;
            int* s_idx = reinterpret_cast<int*>(s_score + BM * WN);
            const int tile_n_id = n0 / BN;
            const float inv_t = tail_inv_temperature(ft.temperature);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mrow = (wm * TM + i) * 16 + r16;  // row inside the tile
                const int m = m0 + mrow;
                float best = -INFINITY;
                int best_i = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nn = n0 + (wn * TN + j) * 16 + kq * 4;
                    if (m < g.M && nn < g.N) {
                        const f32x4 v = epilogue_apply(g.ep, g.N, m, nn, acc[i][j]);
                        if (ft.mode == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (v[e] > best) { best = v[e]; best_i = nn + e; }
                        } else {
                            float lq[4];
                            if constexpr (TAIL_AHEAD) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) lq[e] = tlq[i * TN + j][e];  // (drawn ahead of the main loop: tail_draw)
                            } else {
                                uint32_t rb[4];
                                philox4x32(seed, (uint64_t)(m + row_off) * L4 + (nn >> 2), ft.offset, rb);
#pragma unroll
                                for (int e = 0; e < 4; ++e) lq[e] = log_exp1(rb[e]);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                // labels are visited in increasing order inside a lane, so "first index wins ties" is a strict comparison here (one compare + two
                                // selects per logit instead of the general rule's three compares); the cross-lane merges below keep the general rule
                                const float sc = tail_score_gumbel(v[e], inv_t, lq[e]);
                                if (sc > best) { best = sc; best_i = nn + e; }
                            }
                        }
                    }
                }
#pragma unroll
                for (int o = 16; o < 64; o <<= 1) {  // the 4 lanes (kq = 0..3) that share row r16
                    const float ov = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(best_i, o, 64);
                    argmax_update(best, best_i, ov, oi);
                }
                if (kq == 0) { s_score[mrow * WN + wn] = best; s_idx[mrow * WN + wn] = best_i; }
            }
            __syncthreads();
            if (tid < BM && m0 + tid < g.M) {
                float best = s_score[tid * WN];
                int best_i = s_idx[tid * WN];
#pragma unroll
                for (int w = 1; w < WN; ++w) argmax_update(best, best_i, s_score[tid * WN + w], s_idx[tid * WN + w]);
                const size_t o = (size_t)(m0 + tid) * p.tiles_n + tile_n_id;
                ft.part_score[o] = best;
                ft.part_idx[o] = best_i;
            }
            __syncthreads();  // the scratch is reused by this workgroup's next tile
            return;
        }
        f32x4 qq[GRN_FIN ? TM : 1][GRN_FIN ? TN : 1];  // ring tiles: per 16-row block, column sums of squares (GRN finished in the epilogue)
        const bool grn_fin = GRN_FIN && g.ep.grn_gx_out != nullptr;  // kernel-uniform
        // one 16-row block of the wave tile.  `it` is the loop variable of a fully unrolled loop -- or, on the 32-block wave tile (256x256), an integral constant: a
        // loop body of that size exceeds the compiler's budget for "#pragma unroll", the accumulators would be indexed at run time and live in scratch.
        // (Measured and NOT kept, round 6: fetching the epilogue's operands ahead of its stores -- column operands once per tile, row operands per 16-row block.  gfx950
        // counts vector loads and stores in ONE counter whose two classes complete out of order, so a load behind a store can only be waited for with vmcnt(0); but
        // four workgroups per CU hide that round trip, and the restructured epilogue measured 0.5-1.4 % SLOWER per image in fp32 and no faster on the one-workgroup-per-CU
        // bf16 tiles: profiles/r06_epilogue_operands_ahead_ab.txt.)
        auto ep_row = [&](auto it) __attribute__((always_inline)) {
            const int i = it;
            const int m = m0 + (wm * TM + i) * 16 + r16;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nn = n0 + (wn * TN + j) * 16 + kq * 4;
                const bool ok = m < g.M && nn < g.N;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    f32x4 a = acc[i][j];
                    if constexpr (APRO == 2) {
                        if (!ln_dir[i]) a = (a - *reinterpret_cast<const f32x4*>(g.ln_wsum + nn) * fr_mu[i]) * fr_rs[i];  // the folded LayerNorm (see ln_row_stats)
                    }
                    v = epilogue_apply<BF>(g.ep, g.N, m, nn, a);
                    epilogue_write(g.ep, g.C, g.ldc, m, nn, v);
                }
                if constexpr (GRN_FIN) {
                    if (grn_fin) {
                        f32x4 q = v * v;
                        q[0] = row16_sum(q[0]); q[1] = row16_sum(q[1]); q[2] = row16_sum(q[2]); q[3] = row16_sum(q[3]);
                        qq[i][j] = q;  // every lane of the 16-row group holds the group's column sums (columns nn .. nn + 3)
                    }
                }
                if (g.ep.sumsq_out) {  // kernel-uniform: per-16-row column sums of squares (GlobalResponseNorm statistics)
                    f32x4 q = v * v;
                    q[0] = row16_sum(q[0]); q[1] = row16_sum(q[1]); q[2] = row16_sum(q[2]); q[3] = row16_sum(q[3]);
                    const int mg = m0 + (wm * TM + i) * 16;
                    if (r16 == 0 && nn < g.N && mg < g.M) *reinterpret_cast<f32x4*>(g.ep.sumsq_out + (size_t)(mg >> 4) * g.N + nn) = q;
                }
                if (g.ep.rowstat_out) {  // kernel-uniform: per-row (sum, centred sum of squares) over this 16-column block (LayerNorm-on-load; gemm_device.h)
                    float rs, rq;
                    rowstat_block(v, rs, rq);
                    const int nb = n0 + (wn * TN + j) * 16;
                    if (kq == 0 && m < g.M && nb < g.N) {
                        float* dstp = g.ep.rowstat_out + ((size_t)m * (g.N >> 4) + (nb >> 4)) * 2;
                        dstp[0] = rs; dstp[1] = rq;
                    }
                }
            }
        };
        if constexpr (TM * TN >= 32) {
            static_for<TM>(ep_row);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) ep_row(i);
        }
        if constexpr (GRN_FIN) {
            if (grn_fin) {
                // GlobalResponseNorm's Gx[sample][column] = sqrt(sum over the sample's rows of value^2) finished HERE (reference src/modules.py:37), plus the
                // sum of Gx over this wave's columns -- the consumer adds grn_np such numbers per sample for mean_k Gx.  The host launches this only
                // with tiles whose rows cover whole samples: rows per sample == 16 (every 16-row block is a sample) or == the tile height.
                const int rps = g.ep.grn_rps;
                const int tile_n_id = n0 / BN;
                auto finish = [&](const f32x4 (&qs)[TN], int sample, bool valid) __attribute__((always_inline)) {
                    float wsum = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int nn = n0 + (wn * TN + j) * 16 + kq * 4;
                        f32x4 gx;
#pragma unroll
                        for (int e = 0; e < 4; ++e) gx[e] = sqrtf(qs[j][e]);
                        if (nn < g.N) {
                            if (r16 == 0 && valid) *reinterpret_cast<f32x4*>(g.ep.grn_gx_out + (size_t)sample * g.N + nn) = gx;
                            wsum += (gx[0] + gx[1]) + (gx[2] + gx[3]);
                        }
                    }
                    wsum += __shfl_xor(wsum, 16, 64);  // the four column quads of the 16-column block(s)
                    wsum += __shfl_xor(wsum, 32, 64);
                    if (r16 == 0 && kq == 0 && valid) g.ep.grn_part_out[(size_t)sample * g.ep.grn_np + tile_n_id * WN + wn] = wsum;
                };
                if (rps == 16) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int mg = m0 + (wm * TM + i) * 16;
                        finish(qq[i], mg >> 4, mg < g.M);
                    }
                } else {  // rps == BM: one sample per tile; add the 16-row blocks of this wave, then the waves stacked along M (fixed order)
                    f32x4 qt[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        qt[j] = qq[0][j];
#pragma unroll
                        for (int i = 1; i < TM; ++i) qt[j] += qq[i][j];
                    }
                    if (WM > 1) {
                        float* scr = smem + FLAG_OFF + 16 + TAIL_FLOATS;
                        if (wm > 0 && r16 == 0) {
#pragma unroll
                            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(scr + ((wm * WN + wn) * TN + j) * 16 + kq * 4) = qt[j];
                        }
                        __syncthreads();
                        if (wm == 0) {
                            for (int w = 1; w < WM; ++w)
#pragma unroll
                                for (int j = 0; j < TN; ++j) qt[j] += *reinterpret_cast<const f32x4*>(scr + ((w * WN + wn) * TN + j) * 16 + kq * 4);
                        }
                        __syncthreads();  // the scratch is reused by this workgroup's next tile
                    }
                    if (wm == 0) finish(qt, m0 / rps, m0 < g.M);
                }
            }
        }
    };

    // ---- end of a segment (the unit range left the tile, or ended): K steps [k0, k1) of `tile` are in the accumulators ----
    constexpr int FR = TM * TN * 64 * 4;                     // floats per wave, fragment order [i][j][lane][4]
    constexpr unsigned SLOT_BYTES = (unsigned)(NW * FR * 4);  // one slab = one tile of fp32 partial sums
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, (int)slab_bytes, 0x00020000);
    auto flush = [&](int tile, int k0, int k1, bool first) {
        int tile_m, tile_n;
        sk_tile_coords<(BM >= 64)>(p, tile, tile_m, tile_n);
        // launder the lane id: everything the flush derives from it (fragment offsets, output rows / columns, masks) would
        // otherwise be hoisted out of the unit loop and held in ~80 VGPRs across the MFMA stream
        int lane = PP ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : lane_k;  // (the ping-pong tile's main loop leaves no register to carry it)
        asm volatile("" : "+v"(lane));
        const int r16 = lane & 15, kq = lane >> 4;
        if (DUAL) { acc[0][0] += acc2; acc2 = f32x4{0.f, 0.f, 0.f, 0.f}; }
        bool finish = true;
        if (!(k0 == 0 && k1 == KT)) {
            // Partial tile.  Slab slots are per workgroup: 2*gid for a segment that starts this workgroup's range ("head"),
            // 2*gid + 1 for a later one (necessarily its last).  Write-through (sc1) stores, every storing wave drains
            // vmcnt(0), ONE lane takes a relaxed agent-scope ticket; the last arriver reads all parts back with sc1 loads
            // (no fences, guide G16 recipe R1) in FIXED part order g_first..g_last.
            const unsigned mybase = (2u * gid + (first ? 0u : 1u)) * SLOT_BYTES + (unsigned)(wave * FR * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsrc,
                                                           mybase + ((i * TN + j) * 64 + lane) * 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* sflag = reinterpret_cast<unsigned*>(smem + FLAG_OFF);
            if (tid == 0) sflag[0] = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            // the workgroups whose ranges intersect this tile's units [tb, tb + KT)
            const unsigned tb = (unsigned)tile * (unsigned)KT;
            const unsigned g_first = sk_owner(p, tb);
            const unsigned g_last = sk_owner(p, tb + (unsigned)KT - 1u);
            finish = sflag[0] == g_last - g_first;
            if (finish) {
                if (tid == 0) __hip_atomic_store(tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
                const unsigned off_first = (2u * g_first + (sk_start(p, g_first) < tb ? 1u : 0u)) * SLOT_BYTES;  // g_first came from an earlier tile -> its tail slot
                const unsigned wbase = (unsigned)(wave * FR * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned fo = wbase + ((i * TN + j) * 64 + lane) * 16;
                        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off_first + fo, 0, 16));
                        unsigned gp = g_first + 1;
                        for (; gp + 3 <= g_last; gp += 4) {  // 4 loads in flight, added in part order
                            const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (2u * gp + 0u) * SLOT_BYTES + fo, 0, 16));
                            const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (2u * gp + 2u) * SLOT_BYTES + fo, 0, 16));
                            const f32x4 a2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (2u * gp + 4u) * SLOT_BYTES + fo, 0, 16));
                            const f32x4 a3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (2u * gp + 6u) * SLOT_BYTES + fo, 0, 16));
                            v += a0; v += a1; v += a2; v += a3;
                        }
                        for (; gp <= g_last; ++gp)
                            v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, 2u * gp * SLOT_BYTES + fo, 0, 16));
                        acc[i][j] = v;
                    }
            }
        }
        tail_cur_tile = tile;
        if (finish) epilogue(tile_m * BM, tile_n * BN, r16, kq);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- the unit stream ----
    // invariant at the top of unit i: unit i is in LDS[slot]; R[(i+1)%PD .. (i+PD-1)%PD] hold units i+1..i+PD-1; R[i%PD] is free
    int loaded = 0;  // units fetched so far; the load cursor stops on the range's last unit (re-reading it hits L1/L2)
    if constexpr (RING > 0) {
        // ===== ring variant: RING LDS stages, RING - 1 units in flight, both operands by LDS-DMA, prologue on the fragments =====
        int sidx[GRN_SIDE ? TM : 1];      // GRN prologues: sample (relative to the tile's first) of this lane's fragment rows
        float rinv[(APRO == 4) ? TM : 1];  // APRO 4: 1 / (mean_k gx[sample][:] + 1e-6) of those samples
        int rinv_tile_m = -1;
        auto enter_tile = [&](int tile) __attribute__((always_inline)) {
            if (!GRN_SIDE) return;
            int tile_m, tile_n;
            sk_tile_coords<(BM >= 64)>(p, tile, tile_m, tile_n);
            const int m0 = tile_m * BM, smp0 = (int)fast_div((unsigned)m0, g.a_rps_div);
#pragma unroll
            for (int i = 0; i < TM; ++i) sidx[i] = (int)fast_div((unsigned)min(m0 + (wm * TM + i) * 16 + r16, g.M - 1), g.a_rps_div) - smp0;
            if (APRO == 4 && tile_m != rinv_tile_m) {
                // mean_k gx of the fragment rows' samples from the producer's per-(column tile, wave column) partial sums: the host guarantees
                // a_rows_per_sample % 16 == 0, so the 16 rows of a fragment belong to ONE sample and the whole wave reduces its grn_np numbers
                // (lane-strided loads, xor butterfly: fixed order, every workgroup gets the same bits)
                rinv_tile_m = tile_m;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int smp = __builtin_amdgcn_readfirstlane(smp0 + sidx[i]);
                    const float* sp = g.grn_part + (size_t)smp * g.grn_np;
                    float sm = 0.f;
                    for (int j = lane_k; j < g.grn_np; j += 64) sm += sp[j];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
                    rinv[i] = 1.0f / (sm / (float)g.K + 1e-6f);
                }
            }
        };
        auto compute_ring = [&](int cs, int kt_ring) __attribute__((always_inline)) {
            const float* As = smem + cs * STAGE_FLOATS;
            const float* Bs = As + BM * BK;
            f32x4 af[KG][TM], bf[KG][TN];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                const int c4 = kk * 4 + kq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 16 + r16;
                    af[kk][i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 16 + r16;
                    bf[kk][j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
            }
            if (APRO == 2 && ln_any) {  // (K % 32 == 0 on ring tiles: no K tail to mask; the bf16 form needs the fragment's K position)
#pragma unroll
                for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i) ln_fix(af[kk][i], i, BF_FIX ? kt_ring * BK + (kk * 4 + kq) * 4 : 0);
            }
            if (APRO == 1) {  // GlobalResponseNorm apply on the fragments: a' = a * scale[sample][k] + shift[k] (same expression as the staged form)
                const float* X = Bs + BN * BK;
#pragma unroll
                for (int kk = 0; kk < KG; ++kk) {
                    const int c4 = kk * 4 + kq;
                    const f32x4 t = *reinterpret_cast<const f32x4*>(X + c4 * 4);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(X + 256 + sidx[i] * 32 + c4 * 4);
                        af[kk][i] = af[kk][i] * sc + t;
                    }
                }
            }
            if (APRO == 4) {  // a' = a * (1 + gamma * gx * rinv) + shift  (reference src/modules.py:36-40: gamma * (x * nx) + beta + x, nx = gx / (mean gx + 1e-6))
                const float* X = Bs + BN * BK;
#pragma unroll
                for (int kk = 0; kk < KG; ++kk) {
                    const int c4 = kk * 4 + kq;
                    const f32x4 t = *reinterpret_cast<const f32x4*>(X + c4 * 4);
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(X + 32 + c4 * 4);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const f32x4 gx = *reinterpret_cast<const f32x4*>(X + 64 + sidx[i] * 32 + c4 * 4);
                        const f32x4 sc = gm * gx * rinv[i] + 1.0f;
                        af[kk][i] = af[kk][i] * sc + t;
                    }
                }
            }
            if constexpr (BF) {
                if (TM * TN == 1) {
#pragma unroll
                    for (int kk = 0; kk < KG; kk += 2) {
                        acc[0][0] = mma_bf16(bf[kk][0], af[kk][0], acc[0][0]);
                        acc2 = mma_bf16(bf[kk + 1][0], af[kk + 1][0], acc2);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc[i][j] = mma_bf16(bf[kk][j], af[kk][i], acc[i][j]);
                }
            } else if (TM * TN == 1) {
#pragma unroll
                for (int kk = 0; kk < KG; kk += 2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kk][0][e], af[kk][0][e], acc[0][0], 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kk + 1][0][e], af[kk + 1][0][e], acc2, 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kk][j][e], af[kk][i][e], acc[i][j], 0, 0, 0);
            }
        };
        auto fetch_ring = [&](int stage) __attribute__((always_inline)) {
            load_unit(R[0], stage);
            if (++loaded < n) {  // the cursor stops on the range's last unit: re-issuing it keeps the per-wave DMA count per unit constant (vmcnt arithmetic)
                if (++lkt == KT) {
                    lkt = 0;
                    ++ltile;
                    set_tile(ltile);
                }
            }
        };
        constexpr int PER_UNIT = LA + LB;  // LDS-DMA instructions per unit and wave; the last wave issues 2 (APRO 1) / 3 (APRO 4) more with the GRN side stage
        constexpr int SIDE_DMAS = (APRO == 4 || BIG) ? 3 : 2;
        if constexpr (PP) {
            // ===== 256x256 "ping-pong" tile (tile id 37; bf16 operands): 2 x 4 waves of 128x64 wave tiles, one workgroup per CU, two 64 KiB LDS buffers =====
            // At bf16 MFMA rates (a 16x16x32 MFMA issues in ~17 cycles) the 256x128 tile's K step -- 6 LDS-DMA pieces (60-185 issue cycles each) and 16 fragment
            // reads per wave against 32 MFMAs -- is bound by everything BUT the matrix cores (0.29 busy: profiles/r05_pmc_mfma_busy_bf16_fastmode_config3.txt).
            // This tile halves both per MFMA (128x64 wave tiles: 24 reads and 8 pieces per 64 MFMAs) and hides them behind the OTHER wave of the SIMD, after
            // the MI355X guide's 8-phase template (cdna_hip_programming.md, "The 256^2 8-phase template"):
            //  * a K step (64 bf16) is four PHASES of 16 MFMAs = one quadrant (4 x 2 blocks) of the wave tile x both k groups:
            //        P1 reads A rows 0..63 (8 reads) + W rows 0..31 (4) | P2 reads W rows 32..63 (4) | P3 reads A rows 64..127 (8) | P4 reads nothing (W rows 0..31 kept)
            //  * a phase = [fragment reads, 2 LDS-DMA pieces] barrier [lgkmcnt(0), 16 MFMAs at raised priority] barrier; the wm = 1 waves run ONE barrier behind the
            //    wm = 0 waves (waves w and w + 4 share a SIMD), so one group's MFMA block covers the other group's reads and DMA issue;
            //  * the operand stream is cut into SECTIONS of two pieces per wave, issued one per phase, six sections ahead of their first use:
            //        [A rows 0..63 | 128..191] [W rows 0..127] [W rows 128..255] [A rows 64..127 | 192..255]      (what P1 of each group needs comes first)
            //    Section q = phase + 6 overwrites the buffer half whose last reads (two K steps earlier) were retired by an lgkmcnt(0) at least one barrier before -- for
            //    BOTH groups (W halves: last read in P2, restaged from P4; A rows 0..63 / 128..191: P1, restaged from P3; the other A rows: P3, restaged from the next P2);
            //  * counted waits only: vmcnt(6) in P4 (this wave's pieces of the next K step's first three sections have landed) and vmcnt(8) in P2 (the fourth section
            //    of THIS K step, read from P3 on), each followed by both groups' barriers before the first read of those bytes.  Nothing waits for vmcnt(0) in the loop.
            // Same k order per accumulator as every other bf16 tile (bit-identical results), same flush / epilogue / stream-K decomposition.
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            using I4 = std::integral_constant<int, 4>;
            int lbuf = 0;  // LDS buffer of the load cursor's K step
            auto pp_issue = [&](auto sec_tag) __attribute__((always_inline)) {
                constexpr int sec = decltype(sec_tag)::value;
                float* dAs = smem + lbuf * STAGE_FLOATS + (wave * (64 / SL)) * BK;  // this wave's 8 rows of every 64-row pass (wave-uniform -> M0)
                float* dBs = dAs + BM * BK;
                const int kofs = lkt * (BK * 4);
                if constexpr (sec == 0) {
                    dma_b128_to_lds(rsrcA, dAs, aoff[0], kofs);
                    dma_b128_to_lds(rsrcA, dAs + 2 * RP * BK, aoff[2], kofs);
                } else if constexpr (sec == 1) {
                    dma_b128_to_lds(rsrcW, dBs, boff[0], kofs);
                    dma_b128_to_lds(rsrcW, dBs + RP * BK, boff[1], kofs);
                } else if constexpr (sec == 2) {
                    dma_b128_to_lds(rsrcW, dBs + 2 * RP * BK, boff[2], kofs);
                    dma_b128_to_lds(rsrcW, dBs + 3 * RP * BK, boff[3], kofs);
                } else {
                    dma_b128_to_lds(rsrcA, dAs + RP * BK, aoff[1], kofs);
                    dma_b128_to_lds(rsrcA, dAs + 3 * RP * BK, aoff[3], kofs);
                    lbuf ^= 1;
                    if (++loaded < n) {  // the cursor stops on the range's last K step: re-issuing it keeps the piece count per phase constant (vmcnt arithmetic)
                        if (++lkt == KT) {
                            lkt = 0;
                            ++ltile;
                            set_tile(ltile);
                        }
                    }
                }
            };
            f32x4 PA[2][4], PBl[2][2], PBh[2][2];  // fragments by k group: four A row blocks, the low / high pair of W row blocks
            auto rdA = [&](int st, auto ib_tag) __attribute__((always_inline)) {
                constexpr int ib = decltype(ib_tag)::value;
                const float* As = smem + st * STAGE_FLOATS;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int c4 = kk * 4 + kq;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = (wm * TM + ib + i) * 16 + r16;
                        PA[kk][i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                    }
                }
            };
            auto rdB = [&](f32x4 (&dst)[2][2], int st, auto jb_tag) __attribute__((always_inline)) {
                constexpr int jb = decltype(jb_tag)::value;
                const float* Bs = smem + st * STAGE_FLOATS + BM * BK;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int c4 = kk * 4 + kq;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int row = (wn * TN + jb + j) * 16 + r16;
                        dst[kk][j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                    }
                }
            };
            auto mm = [&](auto ib_tag, auto jb_tag, const f32x4 (&pb)[2][2]) __attribute__((always_inline)) {
                constexpr int ib = decltype(ib_tag)::value, jb = decltype(jb_tag)::value;
                // the compute half of a phase: everything this wave read has arrived, then 16 MFMAs (k group outermost: two MFMAs on one accumulator are 8 apart)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[ib + i][jb + j] = mma_bf16(pb[kk][j], PA[kk][i], acc[ib + i][jb + j]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            };
            // prologue: the first K step whole + the first two sections of the second one
            pp_issue(I0{}); pp_issue(I1{}); pp_issue(I2{}); pp_issue(I3{}); pp_issue(I0{}); pp_issue(I1{});
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // sections 0..2 of the first K step (this wave's pieces)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int cbuf = 0;
            int ctile = (int)fast_div(u0, p.dKT);
            int ckt = (int)(u0 - (unsigned)ctile * (unsigned)KT);
            bool first_seg = true;
            for (int i = 0; i < n;) {
                const int seg_len = min(KT - ckt, n - i);
                if (wm == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here ...
                asm volatile("" ::: "memory");
                for (int s2 = 0; s2 < seg_len; ++s2) {
                    // P1
                    rdA(cbuf, I0{});
                    rdB(PBl, cbuf, I0{});
                    __builtin_amdgcn_sched_barrier(0);
                    pp_issue(I2{});
                    mm(I0{}, I0{}, PBl);
                    // P2
                    rdB(PBh, cbuf, I2{});
                    __builtin_amdgcn_sched_barrier(0);
                    pp_issue(I3{});
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this K step's last section (A rows 64..127 / 192..255), read from P3 on
                    mm(I0{}, I2{}, PBh);
                    // P3
                    rdA(cbuf, I4{});
                    __builtin_amdgcn_sched_barrier(0);
                    pp_issue(I0{});
                    mm(I4{}, I2{}, PBh);
                    // P4
                    pp_issue(I1{});
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // the next K step's first three sections, read from its P1 on
                    mm(I4{}, I0{}, PBl);
                    cbuf ^= 1;
                }
                if (wm == 0) __builtin_amdgcn_s_barrier();  // ... to here: both groups aligned again for the flush (its ticket hand-off needs every wave's stores behind ONE barrier)
                asm volatile("" ::: "memory");
                if (i + seg_len == n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may still be landing when the workgroup's LDS is released
                // LayerNorm row statistics (always the pre-pass form on this tile: one 16-byte load per fragment row) are fetched HERE, not ahead of the unit
                // stream: 40 registers held across the main loop spill (608 bytes per lane); a LayerNorm-consuming range never leaves its tile row
                if (first_seg) ln_row_stats();
                flush(ctile, ckt, ckt + seg_len, first_seg);
                first_seg = false;
                i += seg_len;
                ckt += seg_len;
                if (ckt == KT) { ckt = 0; ++ctile; }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < RING - 1; ++j) fetch_ring(j);
        if constexpr (!BIG) ln_row_stats();  // while the first units are in flight (the 8-wave tile: before its flush -- 16 more registers across its main loop spill)
        int cs = 0, ls = RING - 1;
        int ctile = (int)fast_div(u0, p.dKT);
        int ckt = (int)(u0 - (unsigned)ctile * (unsigned)KT);
        bool first_seg = true;
        if constexpr (BIG && !PP) {
            // ===== 8 waves x (64x64 wave tile): fragments by 16-wide k group, read one group ahead of the MFMAs =====
            //   barrier(u) | DMA unit u+2 -> the stage unit u-1 left | read group 0 of unit u | MFMA group 1 of unit u-1 | read group 1 of unit u | MFMA group 0 of unit u
            // Every ds_read runs under 64 MFMAs that do not need it (the GRN scale / shift fragments ride with the operand fragments and are applied right
            // before the group multiplies); the matrix core only waits at the first unit of a tile.
            f32x4 Sg[APRO == 1 ? KG : 1][APRO == 1 ? TM : 1], Tg[APRO == 1 ? KG : 1];
            auto big_read = [&](auto kk_tag, int st) __attribute__((always_inline)) {
                constexpr int kk = decltype(kk_tag)::value;
                const float* As = smem + st * STAGE_FLOATS;
                const float* Bs = As + BM * BK;
                const int c4 = kk * 4 + kq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 16 + r16;
                    Fa[kk][i] = *reinterpret_cast<const f32x4*>(As + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 16 + r16;
                    Fb[kk][j] = *reinterpret_cast<const f32x4*>(Bs + row * BK + ((c4 ^ (row & (SL - 1))) << 2));
                }
                if constexpr (APRO == 1) {
                    const float* X = Bs + BN * BK;
                    Tg[kk] = *reinterpret_cast<const f32x4*>(X + c4 * 4);
#pragma unroll
                    for (int i = 0; i < TM; ++i) Sg[kk][i] = *reinterpret_cast<const f32x4*>(X + 64 + sidx[i] * 32 + c4 * 4);
                }
            };
            auto big_xform = [&](auto kk_tag, int kt_ring) __attribute__((always_inline)) {  // GlobalResponseNorm apply on the fragments: a' = a * scale[sample][k] + shift[k]
                constexpr int kk = decltype(kk_tag)::value;
                if constexpr (APRO == 1) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) Fa[kk][i] = Fa[kk][i] * Sg[kk][i] + Tg[kk];
                }
                if (APRO == 2 && ln_any) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) ln_fix(Fa[kk][i], i, 0);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I4 = std::integral_constant<int, 4>;
            static_assert(KG == 2, "two k groups per K step");
            // which of the two waves of a SIMD runs the late order: waves w and w + 4 share a SIMD (round-robin placement); p.stagger 2 = by wave parity, 0 = off (A/B)
            const bool late = p.stagger == 1 ? wave >= NW / 2 : (p.stagger == 2 ? (wave & 1) != 0 : false);
            for (int i = 0; i < n;) {
                const int seg_len = min(KT - ckt, n - i);
                enter_tile(ctile);
                bool pend = false;  // group 1 of the previous unit is still to be multiplied
                for (int s2 = 0; s2 < seg_len; ++s2) {
                    // this wave's share of unit u has landed and its reads of the stage that is about to be overwritten (issued two MFMA groups ago) are done
                    if (GRN_SIDE && wave == NW - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 2) * (PER_UNIT + SIDE_DMAS)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 2) * PER_UNIT) : "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    // The two waves that share a SIMD come out of the barrier together.  If both then issue their 6 LDS-DMA pieces (60-185 cycles of issue each,
                    // MI355X_MICROARCH.md) and 8 fragment reads, the matrix core idles for ~10 % of the unit.  STAGGER: one of them multiplies the pending k group
                    // first and does its non-MFMA work while the other one is in its MFMA block, and vice versa.
                    if (!late) {
                        fetch_ring(ls);
                        __builtin_amdgcn_sched_barrier(0);
                        big_read(I0{}, cs);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (pend) {
                        big_xform(I1{}, ckt + s2 - 1);  // k group 1 of the PREVIOUS unit
                        mfma_group(I1{}, I0{}, I4{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (late) {
                        fetch_ring(ls);
                        __builtin_amdgcn_sched_barrier(0);
                        big_read(I0{}, cs);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    big_read(I1{}, cs);
                    __builtin_amdgcn_sched_barrier(0);
                    big_xform(I0{}, ckt + s2);
                    mfma_group(I0{}, I0{}, I4{});
                    __builtin_amdgcn_sched_barrier(0);
                    pend = true;
                    cs = cs + 1 == RING ? 0 : cs + 1;
                    ls = ls + 1 == RING ? 0 : ls + 1;
                }
                big_xform(I1{}, ckt + seg_len - 1);
                mfma_group(I1{}, I0{}, I4{});
                __builtin_amdgcn_sched_barrier(0);
                if (i + seg_len == n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may still be landing when the workgroup's LDS is released
                if (first_seg) ln_row_stats();  // (a LayerNorm-consuming range never leaves its tile row: once)
                flush(ctile, ckt, ckt + seg_len, first_seg);
                first_seg = false;
                i += seg_len;
                ckt += seg_len;
                if (ckt == KT) { ckt = 0; ++ctile; }
            }
            return;
        }
        for (int i = 0; i < n;) {
            const int seg_len = min(KT - ckt, n - i);
            enter_tile(ctile);
            for (int s2 = 0; s2 < seg_len; ++s2) {
                // this wave's share of the oldest unit has landed; the barrier then publishes every wave's share -- and guarantees that all
                // waves are done reading stage `ls` (the unit computed one iteration ago), which the next DMA overwrites
                if (GRN_SIDE && wave == NW - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * (PER_UNIT + SIDE_DMAS)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * PER_UNIT) : "memory");
                // a bare s_barrier: __syncthreads() carries a workgroup-scope fence, which drains EVERY LDS-DMA in flight (vmcnt(0)) and would
                // collapse the ring to one unit.  The fragment reads of the previous unit were consumed by its MFMAs, so nothing else is pending.
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                fetch_ring(ls);
                __builtin_amdgcn_sched_barrier(0);
                compute_ring(cs, ckt + s2);
                __builtin_amdgcn_sched_barrier(0);
                cs = cs + 1 == RING ? 0 : cs + 1;
                ls = ls + 1 == RING ? 0 : ls + 1;
            }
            if (i + seg_len == n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may still be landing when the workgroup's LDS is released
            flush(ctile, ckt, ckt + seg_len, first_seg);
            first_seg = false;
            i += seg_len;
            ckt += seg_len;
            if (ckt == KT) { ckt = 0; ++ctile; }
        }
        return;
    }
    auto fetch = [&](Stage& r, int dma_slot) __attribute__((always_inline)) {
        load_unit(r, dma_slot);
        if (++loaded < n) {  // workgroup-uniform; never runs into the next workgroup's units
            if (APRO == 3 && (lc0 += BK) == g.cv.C) { lc0 = 0; ++ltap; }
            if (++lkt == KT) {
                lkt = 0;
                ltap = 0;
                lc0 = 0;
                ++ltile;
                set_tile(ltile);
            }
        }
    };
    const int tile_first = (int)fast_div(u0, p.dKT);
#pragma unroll
    for (int j = 0; j < PD; ++j) fetch(R[j], 0);  // (DMA variants are 1-deep: unit 0 goes straight to LDS stage 0)
    ln_row_stats();  // while the first units are in flight
    if constexpr (TAIL_AHEAD) tail_draw(tile_first);  // fused tail: the tile's noise, under the same latency (the product's head tiles; the A/B-only head tiles, at their register
                                                       // limit, draw it in the epilogue as before -- same numbers either way)
    store_unit(R[0], 0);
    __syncthreads();
    if constexpr (PIPE) read_group(std::integral_constant<int, 0>{}, 0, (int)(u0 - fast_div(u0, p.dKT) * (unsigned)KT));

    int ctile = (int)fast_div(u0, p.dKT);
    int ckt = (int)(u0 - (unsigned)ctile * (unsigned)KT);
    bool first_seg = true;
    int slot = 0;
    // one unit: `rf` is the free ring stage (gets the unit PD ahead), `rs` holds the next unit (goes to the other LDS stage).
    // One basic block per phase: the prefetch is pinned above the MFMA block (hipcc sinks it otherwise).
    auto step = [&](Stage& rf, const Stage& rs, auto sl_tag) __attribute__((always_inline)) {
        constexpr int sl = decltype(sl_tag)::value;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        fetch(rf, sl ^ 1);  // DMA operands of the next unit start landing in the other LDS stage now (last read before the previous barrier)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIPE) {
            read_group(I1{}, sl, kt_cur);  // k group 1 of this unit; group 0 was read behind the previous barrier
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(I0{}, I0{}, I4{});
            mfma_group(I1{}, I0{}, I2{});
            __builtin_amdgcn_sched_barrier(0);
            store_unit(rs, sl ^ 1);
            __syncthreads();
            read_group(I0{}, sl ^ 1, kt_cur + 1);  // k group 0 of the next unit (a LayerNorm-consuming launch never leaves its tile: the next K step)
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(I1{}, I2{}, I4{});
            __builtin_amdgcn_sched_barrier(0);
        } else {
            compute(sl);
            __builtin_amdgcn_sched_barrier(0);
            store_unit(rs, sl ^ 1);
            __syncthreads();
        }
        ++kt_cur;
    };
    // A segment's units with COMPILE-TIME LDS stages and ring roles (immediate LDS offsets, statically indexed register stages):
    // two instantiations, by the LDS stage the segment starts in.  A segment of odd length leaves the ring in phase 1, so it is
    // re-based with ONE stage copy per segment.
    auto run_segment = [&](auto s0_tag, int len) __attribute__((always_inline)) {
        constexpr int S0 = decltype(s0_tag)::value;
        int s = 0;
        if constexpr (PD == 1) {
            for (; s + 2 <= len; s += 2) {
                step(R[0], R[0], std::integral_constant<int, S0>{});
                step(R[0], R[0], std::integral_constant<int, S0 ^ 1>{});
            }
            if (s < len) step(R[0], R[0], std::integral_constant<int, S0>{});
        } else {
            for (; s + 2 <= len; s += 2) {
                step(R[0], R[1], std::integral_constant<int, S0>{});
                step(R[1], R[0], std::integral_constant<int, S0 ^ 1>{});
            }
            if (s < len) {
                step(R[0], R[1], std::integral_constant<int, S0>{});
                R[1] = R[0];
            }
        }
    };
    for (int i = 0; i < n;) {
        const int seg_len = min(KT - ckt, n - i);
        kt_cur = ckt;
        if (slot == 0) run_segment(std::integral_constant<int, 0>{}, seg_len);
        else run_segment(std::integral_constant<int, 1>{}, seg_len);
        slot ^= seg_len & 1;
        flush(ctile, ckt, ckt + seg_len, first_seg);
        first_seg = false;
        i += seg_len;
        ckt += seg_len;
        if (ckt == KT) { ckt = 0; ++ctile; }
    }
#ifdef PAELLA_GEMM_CLOCK_PROBE
    if (gid == (G >> 1) && tid == 0) {  // a workgroup from the middle of the launch
        g_clock_probe[0] = __builtin_readcyclecounter() - cp_t0;
        g_clock_probe[1] = wall_clock64() - cp_w0;
    }
#endif
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct TileCfg { int wm, wn, tm, tn, pd, bk, ring; };
// BM = wm*tm*16, BN = wn*tn*16; ids are stable (tests and tools name them)
static const TileCfg kCfgs[] = {
    {2, 2, 4, 4, 1, 32},  // 0: 128x128
    {2, 2, 4, 2, 2, 32},  // 1: 128x64
    {2, 2, 2, 2, 2, 32},  // 2: 64x64
    {2, 2, 2, 1, 2, 32},  // 3: 64x32
    {2, 2, 1, 2, 2, 32},  // 4: 32x64
    {2, 2, 1, 1, 2, 32},  // 5: 32x32
    {1, 4, 1, 1, 2, 32},  // 6: 16x64
    {1, 4, 1, 2, 2, 32},  // 7: 16x128
    {1, 4, 2, 2, 2, 32},  // 8: 32x128
    {4, 2, 2, 4, 1, 32},  // 9: 128x128, 8 waves (32x64 wave tiles, fragment-first): the large-problem default
    {2, 4, 4, 2, 1, 32},  // 10: 128x128, 8 waves (64x32 wave tiles, fragment-first)
    {4, 1, 2, 2, 2, 32},  // 11: 128x32, waves stacked along M (32x32 wave tiles): M-covering for M <= 128
    {4, 1, 2, 4, 2, 32},  // 12: 128x64, waves stacked along M (32x64 wave tiles)
    {4, 1, 1, 4, 2, 32},  // 13: 64x64, waves stacked along M (16x64 wave tiles): M-covering for M <= 64
    {4, 2, 2, 2, 2, 32},  // 14: 128x64, 8 waves (32x32 wave tiles)
    {8, 1, 2, 2, 2, 32},  // 15: 256x32, 8 waves stacked along M
    {8, 1, 2, 4, 1, 32},  // 16: 256x64, 8 waves stacked along M (32x64 wave tiles)
    {8, 1, 1, 4, 2, 32},  // 17: 128x64, 8 waves stacked along M (16x64 wave tiles)
    {2, 2, 2, 2, 1, 32},  // 18: 64x64, 1-deep prefetch
    {2, 2, 1, 1, 1, 32},  // 19: 32x32, 1-deep prefetch
    {4, 1, 2, 4, 1, 32},  // 20: 128x64 stacked, 1-deep prefetch
    {4, 1, 1, 2, 2, 32},  // 21: 64x32, waves stacked along M (16x32 wave tiles)
    {2, 2, 1, 4, 2, 32},  // 22: 32x128 (16x64 wave tiles): M-covering for M <= 32
    {1, 4, 2, 1, 2, 32},  // 23: 32x64, waves along N (32x16 wave tiles)
    {2, 2, 1, 1, 2, 64},  // 24: 32x32, K step 64 (half the barriers / address arithmetic per FLOP)
    {2, 2, 1, 1, 1, 64},  // 25: 32x32, K step 64, 1-deep prefetch
    {2, 2, 2, 2, 1, 64},  // 26: 64x64, K step 64, 1-deep prefetch
    {4, 1, 2, 2, 1, 64},  // 27: 128x32 stacked, K step 64, 1-deep prefetch
    {2, 2, 1, 2, 1, 64},  // 28: 32x64, K step 64, 1-deep prefetch
    {4, 2, 2, 2, 1, 64},  // 29: 128x64, 8 waves, K step 64, 1-deep prefetch
    // ring variants (both operands by LDS-DMA into 3 / 4 stages, prologues applied to the fragments); need K % 32 == 0
    {2, 2, 1, 1, 1, 32, 3},  // 30: 32x32, 3 stages (2 units in flight, 5 workgroups per CU)
    {2, 2, 1, 1, 1, 32, 4},  // 31: 32x32, 4 stages (3 units in flight, 4 workgroups per CU)
    {2, 2, 1, 2, 1, 32, 3},  // 32: 32x64, 3 stages
    {2, 2, 2, 1, 1, 32, 3},  // 33: 64x32, 3 stages
    {2, 2, 2, 2, 1, 32, 3},  // 34: 64x64, 3 stages
    {2, 2, 1, 2, 1, 32, 4},  // 35: 32x64, 4 stages
    // the throughput-regime tile: 8 waves of 64x64 wave tiles, both operands by LDS-DMA into 3 stages, fragments read one k group ahead (BIG in the kernel)
    {4, 2, 4, 4, 1, 32, 3},  // 36: 256x128
    // the bf16 fast mode's throughput tile: 2 x 4 waves of 128x64 wave tiles, two 64 KiB LDS buffers, phases of 16 MFMAs with the two wave groups one barrier apart (PP in the kernel)
    {2, 4, 8, 4, 1, 32, 2},  // 37: 256x256 (bf16 operands only)
};
static const int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
int gemm_num_tile_configs() { return kNumCfgs; }

// tile configs that carry the implicit-convolution variant (ids 2, 5, 9, 10, 14, 18)
template <int WM, int WN, int TM, int TN, int PD, int BK>
static constexpr bool conv_tile() {
    return BK == 32 && ((WM == 2 && WN == 2 && TM == TN && (TM == 1 || TM == 2)) || (WM * WN == 8 && WM * TM == 8 && WN * TN == 8) || (WM == 4 && WN == 2 && TM == 2 && TN == 2));
}
static bool conv_cfg(int cfg) { return cfg == 2 || cfg == 5 || cfg == 9 || cfg == 10 || cfg == 14 || cfg == 18 || cfg == 19; }

// tile configs that carry the direct-to-LDS (DMA) twin: the two large-problem workhorses (ids 10 and 18) and the 1-deep 32x32 tile (id 19)
template <int WM, int WN, int TM, int TN, int PD, int BK>
static constexpr bool dma_tile() {
    return BK == 32 && PD == 1 && ((WM == 2 && WN == 4 && TM == 4 && TN == 2) || (WM == 2 && WN == 2 && TM == 2 && TN == 2) || (WM == 2 && WN == 2 && TM == 1 && TN == 1) ||
                                   (WM == 2 && WN == 2 && TM == 4 && TN == 4));  // (id 0: 128x128 on 4 waves of 64x64 wave tiles, two workgroups per CU)
}
static std::atomic<int> g_gemm_raster_gm{8};  // tile rows per rasterisation group (0 = plain m-fastest); test hook
extern "C" int paella_test_gemm_raster(int gm) { g_gemm_raster_gm = gm; return PAELLA_OK; }
static std::atomic<int> g_gemm_dma{1};  // test hook (test_hooks.h): 0 = always the register-staged kernels
extern "C" int paella_test_gemm_dma(int on) { g_gemm_dma = on != 0; return PAELLA_OK; }

template <int TM, int TN, int RING>
static void launch_ring(const GemmArgs& g, const SkPlan& p, unsigned G, float* slabs, unsigned* tickets, unsigned slab_bytes, hipStream_t st) {
    if (g.grn_gx)
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, TM, TN, 1, 4, false, 32, false, RING>), dim3(G), dim3(256), 0, st, g, p, slabs, tickets, slab_bytes);
    else if (g.a_scale)
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, TM, TN, 1, 1, false, 32, false, RING>), dim3(G), dim3(256), 0, st, g, p, slabs, tickets, slab_bytes);
    else if (g.ln_stats)
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, TM, TN, 1, 2, false, 32, false, RING>), dim3(G), dim3(256), 0, st, g, p, slabs, tickets, slab_bytes);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, TM, TN, 1, 0, false, 32, false, RING>), dim3(G), dim3(256), 0, st, g, p, slabs, tickets, slab_bytes);
}

template <int WM, int WN, int TM, int TN, int PD, int BK>
static void launch_one(const GemmArgs& g, const SkPlan& p, unsigned G, float* slabs, unsigned* tickets, unsigned slab_bytes, hipStream_t st) {
    constexpr int NT = 64 * WM * WN;
    if constexpr (dma_tile<WM, WN, TM, TN, PD, BK>()) {
        constexpr bool plain_only = (WM * WN == 4 && TM * TN == 16);  // (the 4-wave 128x128 twin spills with a prologue's extra operands: plain GEMMs only)
        if constexpr (plain_only) {
            if (g_gemm_dma && g.K % BK == 0 && !g.cv.enabled && !g.a_scale && !g.ln_stats) {
                hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 0, false, BK, true>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
                return;
            }
        } else if (g_gemm_dma && g.K % BK == 0) {  // no K tail: operands without a transform go global -> LDS directly
            if (g.cv.enabled)
                hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 3, false, BK, true>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
            else if (g.a_scale)
                hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 1, false, BK, true>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
            else if (g.ln_stats)
                hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 2, false, BK, true>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
            else
                hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 0, false, BK, true>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
            return;
        }
    }
    if constexpr (conv_tile<WM, WN, TM, TN, PD, BK>()) {
        if (g.cv.enabled) {
            hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 3, false, BK>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
            return;
        }
    }
    if (g.a_scale)
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 1, false, BK>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
    else if (g.ln_stats)
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 2, false, BK>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, PD, 0, false, BK>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
}

// ---- bf16 operands (GemmArgs::A16 / W16; the opt-in fast mode): the all-DMA variants of the same kernel ----
template <int WM, int WN, int TM, int TN, bool DMAv, int RINGv>
static void launch_bf(const GemmArgs& g, const SkPlan& p, unsigned G, float* slabs, unsigned* tickets, unsigned slab_bytes, hipStream_t st) {
    constexpr int NT = 64 * WM * WN;
    if (g.ln_stats)
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, 1, 2, false, 32, DMAv, RINGv, true>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, 1, 0, false, 32, DMAv, RINGv, true>), dim3(G), dim3(NT), 0, st, g, p, slabs, tickets, slab_bytes);
}
// tile configs that carry a bf16 variant: the direct-to-LDS twins (10, 18, 19), the ring tiles (30..36) and the ping-pong tile (37)
static bool bf16_cfg(int cfg) { return cfg == 10 || cfg == 18 || cfg == 19 || (cfg >= 30 && cfg <= 37); }
static bool launch_bf_cfg(int cfg, const GemmArgs& g, const SkPlan& p, unsigned G, float* slabs, unsigned* tickets, unsigned slab_bytes, hipStream_t st) {
    switch (cfg) {
        case 10: launch_bf<2, 4, 4, 2, true, 0>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 18: launch_bf<2, 2, 2, 2, true, 0>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 19: launch_bf<2, 2, 1, 1, true, 0>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 30: launch_bf<2, 2, 1, 1, false, 3>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 31: launch_bf<2, 2, 1, 1, false, 4>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 32: launch_bf<2, 2, 1, 2, false, 3>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 33: launch_bf<2, 2, 2, 1, false, 3>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 34: launch_bf<2, 2, 2, 2, false, 3>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 35: launch_bf<2, 2, 1, 2, false, 4>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 36: launch_bf<4, 2, 4, 4, false, 3>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        case 37: launch_bf<2, 4, 8, 4, false, 2>(g, p, G, slabs, tickets, slab_bytes, st); return true;
        default: return false;
    }
}

// ring tiles: both operands by LDS-DMA (whole K steps only), GRN side stage holds the scale rows of at most 8 consecutive samples
static bool ring_ok(const GemmArgs& g, int BM) {
    if (g.K % 32 || g.cv.enabled) return false;
    const int side_samples = BM >= 256 ? 16 : 8;  // sample rows the GRN side stage holds (the 8-wave 256-row tile: 16)
    if ((g.a_scale || g.grn_gx) && (BM - 1) / (g.a_rows_per_sample > 0 ? g.a_rows_per_sample : 1) + 2 > side_samples) return false;
    return true;
}
static inline long tiles_of_cfg(int c, int M, int N) {
    const int BM = kCfgs[c].wm * kCfgs[c].tm * 16, BN = kCfgs[c].wn * kCfgs[c].tn * 16;
    return (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
}

// Tile / workgroup-count choice, fitted to tools/gemm_tune.py sweeps on MI355X (profiles/r02_gemm_tile_sweep*.txt,
// profiles/r02_gemm_midsize_sweep_by_prologue.txt, profiles/r02_gemm_dma_sweep.txt).
// Returns the tile config and G (number of workgroups = number of contiguous unit ranges).  What the sweeps show:
//  * >= 1024 tiles of 128x128: one tile per workgroup -- 64x64 tiles (cfg 18) for plain operands, the 8-wave 128x128 tile (cfg 10) behind a
//    prologue; direct-to-LDS operands, tiles rasterised in groups of 8 rows: 130-140 TFLOP/s;
//  * fewer 128-tiles but >= 5 GFLOP (the batched mid-size shapes, e.g. 4096x1280x1280, 1024x1280x5120, 16384x640x2560): the same
//    tile as ONE persistent workgroup per CU (G = 256 balanced ranges of (tile, K-step) units) -- best or within 1 % of the best
//    variant on every shape swept, 2-5 % ahead of 64x64 tiles when K is long; short K (<= 768) with >= 1024 64x64 tiles: one of
//    those per workgroup (cfg 18);
//  * 2.4-5 GFLOP (1024x1280x1280): 64x64 tiles on 512 balanced ranges;
//  * skinny batch-1 shapes: 32x32 tiles, ~10 K-steps per workgroup, at most 1280 workgroups = 5 per CU, all resident at once
//    (__launch_bounds__(256, 5) on that instantiation guarantees the registers for it) -- every larger
//    tile lands within 5 % of it (24-27 us for 128x5120x1280): these launches are bound by ramp + combine, not by the tile.
// ring tile used for the skinny batch-1 shapes (30..35; 0 = the register-staged / 1-deep DMA kernels of round 2, kept for A/B).  Default 30 (32x32,
// 3 stages; the LayerNorm-prologue GEMMs take its 4-stage sibling 31).  A/B only through the test hook (test_hooks.h); no environment switches.
static std::atomic<int> g_gemm_ring{30};
extern "C" int paella_test_gemm_ring(int cfg) {
    if (cfg != 0 && (cfg < 30 || cfg > 35)) { paella_set_error("ring tile config must be 0 or 30..35"); return PAELLA_ERR_ARG; }
    g_gemm_ring = cfg;
    return PAELLA_OK;
}
static std::atomic<int> g_gemm_big_stagger{1};  // SkPlan::stagger of the 256x128 tile (A/B through the test hook)
extern "C" int paella_test_gemm_big_stagger(int mode) {
    if (mode < 0 || mode > 2) { paella_set_error("stagger mode must be 0, 1 or 2"); return PAELLA_ERR_ARG; }
    g_gemm_big_stagger = mode;
    return PAELLA_OK;
}
static std::atomic<int> g_grn_fuse{1};  // 0 = always the grn_from_partials finalize launch between the two MLP GEMMs (A/B)
extern "C" int paella_test_grn_fuse(int on) { g_grn_fuse = on != 0; return PAELLA_OK; }
// The MLP pair gelu(h W1^T) -> GRN -> W2 of one ResBlock / FeedForwardBlock can skip the GRN finalize launch when (a) both GEMMs are in the skinny class
// the ring tiles serve, (b) the producer's tile rows cover whole samples: 16 rows per sample -> any ring tile (32x32 here), 64 -> the 64x32 tile.
// Returns the ring tile GEMM1 must use (its grn_np is then (N1 / 32) * 2), or 0.
// Measured in the model (profiles/r03_gemm_by_shape_b1_grn_fused.txt): at 16 rows per sample the pair costs +1.5 us and saves a ~6 us launch; at 64 rows
// per sample the 64x32 producer tile (4 parts per tile to combine instead of 2, cross-wave reduction in the epilogue) costs +7.6 us -- more than the
// launch it removes -- so the model asks with allow_64 = false and keeps the finalize launch there.
int gemm_grn_fused_tile(int M, int C4, int C, int rows_per_sample, bool allow_64) {
    if (!g_grn_fuse.load(std::memory_order_relaxed) || !g_gemm_ring.load(std::memory_order_relaxed)) return 0;
    if ((C & 31) || (C4 & 31) || M % rows_per_sample) return 0;
    const double macs = (double)M * C4 * C;
    const long T64 = (long)((M + 63) / 64) * ((C4 + 63) / 64);
    if (macs >= 1.2e9 || T64 >= 1024) return 0;  // the launch heuristic leaves the skinny class there
    if (rows_per_sample == 16) return 30;
    if (rows_per_sample == 64 && allow_64) return 33;
    return 0;
}

// workgroups of a ring tile that are resident at once (LDS-limited; profiles/r03_gemm_ring_resources.txt)
static long ring_resident(int cfg, int apro) {
    switch (cfg) {
        case 30: return 1280;
        case 31: return apro == 1 ? 768 : 1024;
        case 32: case 33: return apro == 1 ? 768 : 1024;
        default: return apro == 1 ? 512 : 768;  // 34, 35
    }
}

// Per-launch-site workgroup counts of the skinny (ring-tile) class: the rules below are global fits; a site (M, N, K, operand prologue class, bf16) listed here takes
// its own count instead.  Entries come from tools/site_tune.py -- coordinate descent over every distinct site of the batch-1 image INSIDE the captured graph,
// two passes, same box (profiles/r05_site_tune_b1.txt) -- and can be overridden at run time through the test hook (what the tuner itself uses).
struct SiteG { int M, N, K, apro, bf, G, cfg; };  // cfg: 0 = the rule's tile (only G is overridden; skinny ring class), > 0 = this tile id + 1 (any class)
static SiteG g_sites[96] = {
    // profiles/r05_site_tune_b1.txt: 27 skinny sites of the batch-1 image swept, 5 moved (22.47 -> 22.24 ms per image on the tuning box, -1.0 %; the second pass changed
    // nothing); everywhere else the global rules sit within 30 us per image of the best candidate
    {128, 1280, 1280, 0, 0, 768},   // attention out-projection, 64-position level (rule: 640)
    {32, 1280, 5120, 1, 0, 400},    // MLP out with the GRN prologue, 16-position level (rule: 640)
    {32, 5120, 1280, 0, 0, 768},    // MLP in, 16-position level (rule: 640)
    {32, 1280, 1280, 0, 0, 200},    // attention out-projection, 16-position level (rule: 160)
    {1024, 384, 1536, 0, 0, 768},   // VQGAN bottleneck MLP out (rule: 1280)
    // bf16 fast mode (profiles/r05_site_tune_b1_bf16.txt: 24 sites, 1 moved, -0.45 %)
    {32, 1280, 1280, 0, 1, 200},    // attention out-projection, 16-position level (rule: 40)
};
// the built-in entries are counted in a static initialiser (before any launch can race on it); run-time overrides are appended behind them
static int count_builtin_sites() {
    int n = 0;
    while (n < 96 && g_sites[n].M > 0) ++n;
    return n;
}
static const int g_nsites_builtin = count_builtin_sites();
static std::atomic<int> g_nsites{g_nsites_builtin};
static const SiteG* site_find(int M, int N, int K, int apro, int bf) {
    const int n = g_nsites.load(std::memory_order_acquire);
    for (int i = n - 1; i >= 0; --i)  // (later entries -- run-time overrides -- win)
        if (g_sites[i].M == M && g_sites[i].N == N && g_sites[i].K == K && g_sites[i].apro == apro && g_sites[i].bf == bf) return &g_sites[i];
    return nullptr;
}
static int site_lookup(int M, int N, int K, int apro, int bf) {  // workgroup count of a skinny-class site (entries that also name a tile are handled by the caller)
    const SiteG* e = site_find(M, N, K, apro, bf);
    return (e && e->cfg == 0) ? e->G : 0;
}
// test hook: G > 0 sets / overrides a site, G == 0 removes the run-time entries of that site, M == 0 removes every run-time entry
static int site_set(int M, int N, int K, int apro, int bf, int G, int cfg1);
extern "C" int paella_test_gemm_site(int M, int N, int K, int apro, int bf, int G) { return site_set(M, N, K, apro, bf, G, 0); }
// the same with an explicit tile id (any class; what tools/site_tune_mid.py sweeps): cfg < 0 = only the workgroup count
extern "C" int paella_test_gemm_site_cfg(int M, int N, int K, int apro, int bf, int cfg, int G) {
    if (cfg >= kNumCfgs) { paella_set_error("bad tile config %d", cfg); return PAELLA_ERR_ARG; }
    return site_set(M, N, K, apro, bf, G, cfg < 0 ? 0 : cfg + 1);
}
static int site_set(int M, int N, int K, int apro, int bf, int G, int cfg1) {
    int n = g_nsites.load();
    if (M == 0) { g_nsites = g_nsites_builtin; return PAELLA_OK; }
    for (int i = n - 1; i >= g_nsites_builtin; --i)
        if (g_sites[i].M == M && g_sites[i].N == N && g_sites[i].K == K && g_sites[i].apro == apro && g_sites[i].bf == bf) {
            if (G > 0) { g_sites[i].G = G; g_sites[i].cfg = cfg1; return PAELLA_OK; }
            g_sites[i] = g_sites[n - 1]; g_nsites = n - 1; return PAELLA_OK;
        }
    if (G <= 0) return PAELLA_OK;
    if (n >= 96) { paella_set_error("site table full"); return PAELLA_ERR_STATE; }
    g_sites[n] = SiteG{M, N, K, apro, bf, G, cfg1};
    g_nsites.store(n + 1, std::memory_order_release);
    return PAELLA_OK;
}

// test hook (tests/test_kernel_resources.py): the table above, so that a CPU test can hold it against the occupancy the compiler actually produced
extern "C" long paella_test_ring_resident(int cfg, int apro) { return (cfg >= 30 && cfg <= 35) ? ring_resident(cfg, apro) : -1; }

static void choose_config(int M, int N, int K, int apro, bool ring_allowed, int force_ring, size_t slab_cap_bytes, int* cfg_out, unsigned* G_out) {
    // one consistent value per decision; force_ring: the caller needs THIS ring tile in the skinny class (its epilogue finishes GRN per tile)
    const int g_gemm_ring = ::g_gemm_ring.load(std::memory_order_relaxed) ? (force_ring > 0 ? force_ring : ::g_gemm_ring.load(std::memory_order_relaxed)) : 0;
    const long ktiles = (K + 31) / 32;
    const double macs = (double)M * N * K;
    const long T128 = tiles_of_cfg(10, M, N), T64 = tiles_of_cfg(18, M, N), T32 = tiles_of_cfg(5, M, N);
    int cfg;
    long G;
    const SiteG* const site = site_find(M, N, K, apro, 0);
    if (site && site->cfg > 0 && (!kCfgs[site->cfg - 1].ring || ring_allowed)) {
        // a launch site with its own (tile, workgroup count): fitted inside the captured batch-32 graph (tools/site_tune_mid.py, profiles/r06_site_tune_mid_b32.txt)
        cfg = site->cfg - 1;
        G = site->G;
    } else if (T128 >= 1024) {
        // plain operands: 64x64 tiles, 4 independent workgroups per CU, grouped rasterisation (140 TFLOP/s on 32768x5120x1280);
        // with a prologue the 8-wave 128x128 tile stages the A operand half as often and ties or wins
        // (a LayerNorm-consuming GEMM multiplies the raw operand since round 3 -- the normalisation is folded into its epilogue -- and takes the plain rule:
        // configs[2] 19.78 -> 20.41 images/s, profiles/r03_ring_rules_ab.txt)
        if (apro == 0 || apro == 2) { cfg = 18; G = T64; }
        else { cfg = 10; G = T128; }
    } else if (macs >= 2.5e9) {
        if ((K <= 768 || apro == 2) && T64 >= 1024) { cfg = 18; G = T64; }  // (LayerNorm-folded GEMMs: one statistics pass per workgroup -> one tile per workgroup)
        // plain operands: the mid-size sweep INSIDE the batch-32 model (tools/site_tune_mid.py, profiles/r06_site_tune_mid_b32.txt: 18 sites x 4 tiles x workgroup counts)
        // put the 64x64 direct-to-LDS tile 3-8 % ahead of 256 persistent 128x128 workgroups on every plain site of this class -- one tile per workgroup from 2048
        // tiles up, 1024 balanced ranges below (4096x1280x1280 124 -> 116 us, 1024x5120x1280 121 -> 116, 2048x5120x1280 221 -> 211, 8192x640x1024 102 -> 94);
        // with a GRN / LayerNorm prologue the old rule stays within 1 % of the best candidate.  Worth 0.5 % per image at batch 32: the class is closed.
        else if (apro == 0) { cfg = 18; G = T64 >= 2048 ? T64 : 1024; }
        else { cfg = 10; G = 256; }
    } else if (g_gemm_ring && ring_allowed && macs >= 1.2e9 && macs < 2.4e9 && T64 < 1024 && apro != 2) {
        // 1.2-2.4 GFLOP with few tiles (256x5120x1280, 1024x1280x1280, ...): the 32x32 ring tile on every resident slot is 8-12 % ahead of 64x64 tiles
        cfg = g_gemm_ring;
        G = ring_resident(cfg, apro);
        const long Tc = tiles_of_cfg(cfg, M, N);
        if (G < Tc) G = Tc;
    } else if (macs >= 1.2e9 || T64 >= 1024) {
        cfg = 18;
        G = (T64 >= 2048 || ktiles < 16) ? T64 : 512;  // short K: ranges would be mostly partial tiles
    } else {
        // plain operands from 128 rows up: the 1-deep twin whose operands go global -> LDS directly is 2-5 % ahead (profiles/r02_gemm_dma_sweep.txt)
        cfg = (apro == 0 && M >= 128 && K % 32 == 0) ? 19 : 5;
        long resident = (apro == 1 || apro == 2) ? 1024 : 1280;  // workgroups that fit at once (see the launch bounds above)
        if (g_gemm_ring && ring_allowed) {
            // LDS-DMA ring tiles: 5-15 % ahead of the register-staged / 1-deep tiles on every batch-1 shape in isolation (profiles/r03_gemm_ring_sweep.txt)
            // and 5.4 % per image in the model.  Every launch carries ~6-7 us of fixed latency (boundary, first fetch, publish / ticket / combine,
            // epilogue round trips) on top of a K loop that runs at ~130 TFLOP/s.  The workgroup count was fitted IN THE MODEL (profiles/r03_ring_rules_ab.txt:
            // isolated launches with L2-warm activations preferred fewer workgroups and did not predict the model):
            //  * ~10 K-steps per workgroup, up to every resident slot (1280);
            //  * LayerNorm prologue: every workgroup re-derives its rows' statistics from the producer's partials, which costs more than a K split
            //    saves -- one tile per workgroup from 160 tiles up (128x3840x1280: 21.0 us against 25.2 with a 2-way split), 2-way below; 4-stage tile.
            cfg = (g_gemm_ring == 30 && apro == 2) ? 31 : g_gemm_ring;
            resident = ring_resident(cfg, apro);
            const long Tc = tiles_of_cfg(cfg, M, N);
            const long U = Tc * ktiles;
            if (apro == 2) G = Tc >= 160 ? Tc : 2 * Tc;
            else G = U / 10;
            if (const int gs = site_lookup(M, N, K, apro, 0)) G = gs;
            if (G < Tc) G = Tc;
            if (G > resident) G = resident;
        } else {
            const long U = T32 * ktiles;
            G = U / 10;
            if (G < T32) G = T32;
            if (G > resident) G = resident;
        }
    }
    const long T = tiles_of_cfg(cfg, M, N);
    const long U = T * ktiles;
    if (G > U) G = U;
    if (G < 1) G = 1;
    // workspace limits: partial tiles need 2 slab slots per workgroup and one ticket per tile
    const int BM = kCfgs[cfg].wm * kCfgs[cfg].tm * 16, BN = kCfgs[cfg].wn * kCfgs[cfg].tn * 16;
    if (G != T) {
        const size_t slot = (size_t)BM * BN * sizeof(float);
        if (T > (long)kGemmMaxTickets || slot * 2 > slab_cap_bytes) G = T;
        else if ((size_t)G * 2 * slot > slab_cap_bytes) G = (long)(slab_cap_bytes / (2 * slot));
    }
    *cfg_out = cfg;
    *G_out = (unsigned)G;
}

// Tile / workgroup-count choice for bf16 operands (the opt-in fast mode).  The matrix cores run 16x faster on the same LDS bytes, so what a tile costs
// besides its MFMAs -- fragment reads (a 16x16x32 MFMA is 16 cycles against the 4 + 4 LDS cycles of its two ds_read_b128), LDS-DMA issue, barriers, the
// epilogue -- decides: the 256x128 tile (id 36: 64x64 wave tiles = half the fragment bytes per MFMA of the 128x128 tile, a quarter of the 64x64 tile's; reads
// issued one k group ahead) is the throughput tile here although it loses to the 64x64 tile in fp32.  Fitted to tools/gemm_tune.py --bf16 sweeps
// (profiles/r05_gemm_bf16_tile_sweep.txt).  K steps are 64 elements wide.
static std::atomic<int> g_bf16_rule{0};  // test hook (A/B of the rules below): bit 0 = never tile 36 (the fp32 rules' tiles instead)
extern "C" int paella_test_gemm_bf16_rule(int mask) { g_bf16_rule = mask; return PAELLA_OK; }
static void choose_config_bf16(int M, int N, int K, int apro, size_t slab_cap_bytes, int* cfg_out, unsigned* G_out) {
    const long ktiles = K / 64;
    const double macs = (double)M * N * K;
    const long T256 = tiles_of_cfg(36, M, N), T128 = tiles_of_cfg(10, M, N), T64 = tiles_of_cfg(18, M, N), TPP = tiles_of_cfg(37, M, N);
    const int rule = g_bf16_rule.load(std::memory_order_relaxed);
    const bool no_big = (rule & 1) != 0, no_persist = (rule & 2) != 0, no_pp = (rule & 4) != 0;
    int cfg;
    long G;
    // (profiles/r05_gemm_bf16_tile_sweep.txt; TFLOP/s in isolation, fp32 outputs)
    // share of the chip's tile slots a launch of T one-per-workgroup tiles keeps busy over its ceil(T / 256) rounds
    auto round_eff = [](long T) { return (double)T / (double)(((T + 255) / 256) * 256); };
    if (!no_big && !no_pp && K >= 2560 && (N % 256) == 0 && TPP >= 128 && 1.25 * round_eff(TPP) >= round_eff(T256)) {
        // LONG K (the MLP's second GEMM, K = 4c): the 256x256 ping-pong tile, one tile per workgroup.  Its main loop sustains 1.3-1.5 PFLOP/s-equivalent per busy CU
        // (8192x1280x5120 on 160 CUs: 917 TFLOP/s against 764 for the 256x128 tile; 32768x1280x5120, 2.5 rounds of the chip: 1050 against 989), but every tile pays a
        // ramp of ~100 KB of operands per CU and an epilogue of 32 fragments per wave that nothing overlaps (one workgroup per CU): at K = 1280 it is 5-15 % BEHIND the
        // 256x128 tile (32768x5120x1280: 820 against 862; 32768x1280x1280: 674 against 812), below 128 tiles it leaves half the chip idle (4096x1280x5120: 493 against 755),
        // and N that is not a multiple of 256 wastes tile columns (131072x640x2560: 732 against 829).  Half as many tiles also quantise worse into rounds of 256: 320 tiles
        // (16384x1280x5120, batch 128 at 32x32 tokens) fill 62 % of two rounds where 640 tiles of 256x128 fill 83 % of three -- measured -0.4 % per image in the model -- so
        // the rule asks for 1.25 x the round efficiency of the 256x128 launch (the main-loop advantage).  profiles/r06_gemm_bf16_pingpong_tile.txt, r06_bf16_pingpong_rule_ab.txt
        cfg = 37; G = TPP;
    } else if (T256 >= 1024 && (K <= 768 || N < 512)) {
        // short K (level-0 MLP in: 131072x2560x640, the VQGAN's 384- / 192-wide blocks) or few tile columns: IN THE MODEL (bias + GELU + bf16 store + GRN statistics in
        // the epilogue; profiles/r05_gemm_by_shape_bf16_config3_rule_ab.txt) the epilogue is as long as the main loop, and four independent 64x64 workgroups per CU
        // overlap one's epilogue with another's main loop: 131072x2560x640 1 114 us against 1 293 (256x128, one tile per workgroup) and 1 576 (persistent ranges,
        // the winner of the isolated sweep); 262144x1536x384 1 013 against 1 238
        cfg = 18; G = T64;
    } else if (!no_big && T256 >= 1024) {
        // BASELINE configs[2]-class launches: 256x128 tiles.  Long K (>= 2560): one tile per workgroup (32768x1280x5120: 950, 131072x640x2560: 800).  Short K: the
        // epilogue is a large share of a tile, so 256 persistent workgroups walk balanced ranges of >= 4 tiles each and one tile's stores overlap the next one's
        // operand stream (32768x5120x1280: 819 against 636; 131072x2560x640: 500-515 against 465).  A LayerNorm-consuming launch keeps one tile per workgroup
        // (its row statistics are derived once per workgroup).
        cfg = 36;
        G = (K <= 1280 && N >= 2048 && apro != 2 && !no_persist) ? 256 : T256;
    } else if (!no_big && T256 >= 128 && (K >= 2560 || T256 >= 256) && !(K <= 1280 && T64 >= 4096)) {
        cfg = 36; G = T256;    // 4096x1280x5120: 723 (160 tiles); 4096x3840x1280: 604; 16384x640x2560: 603
    } else if (T64 >= 4096) {
        cfg = 18; G = T64;     // 4096x5120x1280: 620 against 563 for 640 tiles of 256x128 (2.5 rounds of the chip)
    } else if (T64 >= 256 && macs < 1.0e10 && K <= 1280) {
        cfg = 34; G = tiles_of_cfg(34, M, N);   // 64x64 ring tile: 4096x1280x1280 456, 1024x5120x1280 394, 1024x1280x1280 268
    } else if (macs >= 2.5e9) {
        cfg = 10; G = T128 >= 256 ? T128 : 256;  // 1024x1280x5120: 318 on 256 balanced ranges of 128x128 tiles
    } else {
        // skinny batch-1 launches: the 4-stage 32x32 ring tile; with the matrix cores 16x faster a launch is its latency chain, and every K split adds a
        // publish / ticket / combine round trip -- one tile per workgroup from 160 tiles up or for K <= 1280, otherwise split towards ~512 workgroups with
        // at least 16 K steps (1024 elements) each (128x5120x1280: 12.0 us unsplit against 14.0 on 1280 ranges; 128x1280x5120: 12.4 on 512 against 16.1)
        cfg = 31;
        const long Tc = tiles_of_cfg(cfg, M, N);
        long S = (512 + Tc / 2) / (Tc > 0 ? Tc : 1);
        if (S > ktiles / 16) S = ktiles / 16;
        if (S < 1 || apro == 2) S = 1;
        G = Tc * S;
        const long resident = ring_resident(cfg, apro);
        if (const int gs = site_lookup(M, N, K, apro, 1)) { G = gs; if (G < Tc) G = Tc; }
        if (G > resident && G > Tc) G = resident;
    }
    const long T = tiles_of_cfg(cfg, M, N);
    const long U = T * ktiles;
    if (G > U) G = U;
    if (G < 1) G = 1;
    const int BM = kCfgs[cfg].wm * kCfgs[cfg].tm * 16, BN = kCfgs[cfg].wn * kCfgs[cfg].tn * 16;
    if (G != T) {
        const size_t slot = (size_t)BM * BN * sizeof(float);
        if (T > (long)kGemmMaxTickets || slot * 2 > slab_cap_bytes) G = T;
        else if ((size_t)G * 2 * slot > slab_cap_bytes) G = (long)(slab_cap_bytes / (2 * slot));
    }
    *cfg_out = cfg;
    *G_out = (unsigned)G;
}

// ---------------------------------------------------------------------------
// optional per-launch timing (HIP events on the launch stream) for bench.py's roofline line
// ---------------------------------------------------------------------------
struct GemmProf {
    bool on = false;
    std::vector<hipEvent_t> pool;   // events, used pairwise
    size_t used = 0;
    std::vector<double> flops, bytes;
    std::vector<int> shape;         // per launch: M, N, K, prologue (0 none, 1 GRN, 2 LayerNorm, 3 implicit conv), tail
};
static GemmProf g_prof;
static std::mutex g_prof_mu;  // enable / record / collect may come from different host threads (one per device in a multi-GPU process)

static int launch_gemm_cfg_impl(const GemmArgs& g, int cfg, int splitk, void* ws, size_t ws_bytes, hipStream_t st);

// every dense-contraction launch -- the head GEMM with the fused tail included -- goes through this bracket
template <typename F>
static int prof_bracket(const GemmArgs& g, hipStream_t st, bool stores_c, F&& launch) {
    if (!g_prof.on) return launch();
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof.used + 2 > g_prof.pool.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            HIP_CHECK_RET(hipEventCreate(&e));
            g_prof.pool.push_back(e);
        }
    }
    hipEvent_t e0 = g_prof.pool[g_prof.used], e1 = g_prof.pool[g_prof.used + 1];
    HIP_CHECK_RET(hipEventRecord(e0, st));
    const int rc = launch();
    HIP_CHECK_RET(hipEventRecord(e1, st));
    g_prof.used += 2;
    g_prof.flops.push_back(2.0 * g.M * g.N * g.K);
    g_prof.bytes.push_back(4.0 * ((double)g.M * g.K + (double)g.N * g.K + (stores_c ? (double)g.M * g.N : 0.0)));
    const int shp[5] = {g.M, g.N, g.K, g.cv.enabled ? 3 : (g.grn_gx ? 4 : (g.a_scale ? 1 : (g.ln_stats ? 2 : 0))), stores_c ? 0 : 1};
    g_prof.shape.insert(g_prof.shape.end(), shp, shp + 5);
    return rc;
}

// threshold of the LayerNorm fold (gemm_device.h: kLnFoldMaxRatio); the test hook moves it to measure the fold's error curve (inf = always fold, 0 = never)
static std::atomic<float> g_ln_fold_ratio{kLnFoldMaxRatio};
extern "C" int paella_test_ln_fold_ratio(float ratio) { g_ln_fold_ratio = ratio; return PAELLA_OK; }
// test hook: a device word that counts the waves of LayerNorm-consuming GEMM launches whose rows took the operand-side path (null = off, the default)
static std::atomic<unsigned*> g_ln_guard_count{nullptr};
extern "C" int paella_test_ln_guard_counter(unsigned* dev_word) { g_ln_guard_count = dev_word; return PAELLA_OK; }

int launch_gemm_cfg(const GemmArgs& g_in, int cfg, int splitk, void* ws, size_t ws_bytes, hipStream_t st) {
    GemmArgs g = g_in;
    g.ln_fold_ratio = g_ln_fold_ratio.load(std::memory_order_relaxed);
    g.ln_guard_count = g_ln_guard_count.load(std::memory_order_relaxed);
    g.a_rps_div = fast_div_of((unsigned)(g.a_rows_per_sample > 0 ? g.a_rows_per_sample : 1));
    g.ep.rps_div = fast_div_of((unsigned)(g.ep.rows_per_sample > 0 ? g.ep.rows_per_sample : 1));
    return prof_bracket(g, st, true, [&]() { return launch_gemm_cfg_impl(g, cfg, splitk, ws, ws_bytes, st); });
}

extern "C" int paella_prof_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.on = on != 0;
    g_prof.used = 0;
    g_prof.flops.clear();
    g_prof.bytes.clear();
    g_prof.shape.clear();
    return PAELLA_OK;
}

// Per-launch records since paella_prof_enable(1), WITHOUT resetting them (call before paella_prof_collect): us_out[i] = duration of launch i,
// shape_out[5 i ..] = M, N, K, prologue, fused-tail flag.  Returns the number of launches (at most cap are written).
extern "C" long long paella_prof_detail(float* us_out, int* shape_out, long long cap) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    const size_t n = g_prof.used / 2;
    for (size_t i = 0; i < n && (long long)i < cap; ++i) {
        if (hipEventSynchronize(g_prof.pool[2 * i + 1]) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.pool[2 * i], g_prof.pool[2 * i + 1]) != hipSuccess) return -1;
        us_out[i] = t * 1e3f;
        for (int k = 0; k < 5; ++k) shape_out[5 * i + k] = g_prof.shape[5 * i + k];
    }
    return (long long)n;
}

// Sums the event-timed GEMM launches recorded since paella_prof_enable(1) (synchronises on the recorded events).
extern "C" int paella_prof_collect(double* total_ms, double* total_flops, double* total_bytes, int64_t* launches) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    double ms = 0, fl = 0, by = 0;
    const size_t n = g_prof.used / 2;
    for (size_t i = 0; i < n; ++i) {
        HIP_CHECK_RET(hipEventSynchronize(g_prof.pool[2 * i + 1]));
        float t = 0.f;
        HIP_CHECK_RET(hipEventElapsedTime(&t, g_prof.pool[2 * i], g_prof.pool[2 * i + 1]));
        ms += t; fl += g_prof.flops[i]; by += g_prof.bytes[i];
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    if (launches) *launches = (int64_t)n;
    g_prof.used = 0;
    g_prof.flops.clear();
    g_prof.bytes.clear();
    g_prof.shape.clear();
    return PAELLA_OK;
}

// `ws` = a split-K region: kGemmTicketBytes of tickets (zero when first handed over, see paella_workspace_init) followed
// by slab space.  cfg < 0: heuristic.  splitk > 0: G = tiles * splitk (classic split-K); splitk < 0: G = -splitk workgroups.
static const int kLnPrepassMinRows = 2048;  // from here up the LayerNorm row statistics are finished by one small launch instead of by every workgroup
static int launch_gemm_cfg_impl(const GemmArgs& g_in, int cfg, int splitk, void* ws, size_t ws_bytes, hipStream_t st) {
    GemmArgs g = g_in;
    if (g.M <= 0 || g.N <= 0) return PAELLA_OK;
    if ((g.K & 3) || (g.N & 3) || (g.lda & 3) || (g.ldw & 3) || (g.ldc & 3 && g.ep.store_mode != STORE_PIXSHUF_NCHW)) {
        paella_set_error("gemm: K, N, lda, ldw, ldc must be multiples of 4 (M=%d N=%d K=%d lda=%d ldw=%d ldc=%d)",
                         g.M, g.N, g.K, g.lda, g.ldw, g.ldc);
        return PAELLA_ERR_ARG;
    }
    if ((g.ep.rowstat_out && (g.N & 15)) || (g.ln_stats && (g.a_scale || g.K != 16 * g.ln_nblk || !g.ln_wsum))) {
        paella_set_error("gemm: row statistics need N %% 16 == 0, K == 16 * ln_nblk and the weight's row sums (ln_wsum)");
        return PAELLA_ERR_ARG;
    }
    if (g.ep.store_mode == STORE_D2S && (g.ep.sC & 3)) {
        paella_set_error("gemm: depth-to-space store needs channels %% 4 == 0");
        return PAELLA_ERR_ARG;
    }
    const bool bf = g.A16 != nullptr || g.W16 != nullptr;  // opt-in fast mode: bf16 operands, set per launch by the caller (never by a process-wide switch)
    if (bf && (!g.A16 || !g.W16 || !gemm_bf16_ok(g.K, g.lda, g.ldw) || g.a_scale || g.grn_gx || g.ep.grn_gx_out || g.cv.enabled || (((uintptr_t)g.A16 | (uintptr_t)g.W16) & 15))) {
        paella_set_error("gemm: bf16 operands need A16 and W16 (16-byte aligned), K %% 64 == 0, lda / ldw %% 8 == 0 and no GRN / convolution prologue (M=%d N=%d K=%d lda=%d ldw=%d)",
                         g.M, g.N, g.K, g.lda, g.ldw);
        return PAELLA_ERR_ARG;
    }
    if (g.cv.enabled) {
        if (g.a_scale || g.ln_stats || g.cv.ntaps < 1 || g.cv.ntaps > 16 || (g.cv.C & 31) || g.K != g.cv.ntaps * g.cv.C || g.cv.Ho < 1 || g.cv.Wo < 1 ||
            g.M % (g.cv.Ho * g.cv.Wo)
            || (size_t)2 * g.cv.Hi * g.cv.Wi * g.cv.C * sizeof(float) > 0xffffffffull) {
            paella_set_error("gemm: bad implicit-convolution descriptor (C=%d must be a multiple of 32, K=%d == ntaps*C, M=%d a multiple of Ho*Wo)", g.cv.C, g.K, g.M);
            return PAELLA_ERR_ARG;
        }
    }
    const bool have_ws = ws && ws_bytes > kGemmTicketBytes;
    size_t slab_cap = have_ws ? ws_bytes - kGemmTicketBytes : 0;
    // The LayerNorm row pre-pass: statistics finished once per row.  bf16 operands: the same launch rewrites the rows of the bf16 copy whose |mean| / std exceeds
    // the fold threshold as bf16(LayerNorm(fp32 row)) (the consumer then skips the fold for them: gemm_nt_kernel, ln_pre).
    auto ln_prepass = [&]() -> int {
        // the finished statistics live at the END of the split-K region (the slabs of this launch, if any, start at its front); the carve-out is aligned down
        // whatever ws_bytes the caller passed, and only taken when >= 80 MiB of slab space remain (the largest launch shape, 256 ranges of 256x128 tiles, needs 64)
        const size_t bytes = ((size_t)g.M * 16 + 255) & ~(size_t)255;
        const size_t row4_off = (ws_bytes - bytes) & ~(size_t)255;
        float* row4 = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + row4_off);
        slab_cap = row4_off - kGemmTicketBytes;
        const int rc = launch_ln_rowstat_finalize(g.ln_stats, g.ln_nblk, g.K, g.ln_eps, row4, g.M, bf ? g.A : nullptr, (bf && g.A) ? const_cast<unsigned short*>(g.A16) : nullptr, g.lda,
                                                  g_ln_fold_ratio.load(std::memory_order_relaxed), g_ln_guard_count.load(std::memory_order_relaxed), st);
        if (rc != PAELLA_OK) return rc;
        g.ln_row = row4;
        return PAELLA_OK;
    };
    const bool prepass_fits = slab_cap >= ((size_t)80 << 20) + (size_t)g.M * 16;
    if (g.ln_stats && !g.ln_row && g.M >= kLnPrepassMinRows && prepass_fits) {
        const int rc = ln_prepass();
        if (rc != PAELLA_OK) return rc;
    }
    unsigned G = 0;
    if (cfg < 0 && bf) {
        choose_config_bf16(g.M, g.N, g.K, g.ln_stats ? 2 : 0, slab_cap, &cfg, &G);
    } else if (cfg < 0) {
        choose_config(g.M, g.N, g.K, (g.a_scale || g.grn_gx) ? 1 : (g.ln_stats ? 2 : 0), ring_ok(g, 64), g.force_ring_cfg, slab_cap, &cfg, &G);
        if (g.cv.enabled && !conv_cfg(cfg)) { paella_set_error("internal: heuristic picked tile %d without a convolution variant", cfg); return PAELLA_ERR_STATE; }
    } else {
        if (cfg >= kNumCfgs) { paella_set_error("gemm: bad tile config %d", cfg); return PAELLA_ERR_ARG; }
        const long T = tiles_of_cfg(cfg, g.M, g.N);
        G = splitk < 0 ? (unsigned)(-splitk) : (unsigned)(T * (splitk < 1 ? 1 : splitk));
        if (g.cv.enabled && !conv_cfg(cfg)) { paella_set_error("gemm: tile config %d has no implicit-convolution variant", cfg); return PAELLA_ERR_ARG; }
    }
    const TileCfg& tc = kCfgs[cfg];
    const int BM = tc.wm * tc.tm * 16, BN = tc.wn * tc.tn * 16;
    // the 8-wave bf16 tiles have no in-kernel operand-side guard: the pre-pass it is, whatever M; the ping-pong tile carries ONLY the pre-pass form of the row statistics
    if (bf && g.ln_stats && !g.ln_row && (tc.ring == 2 || (g.A && tc.wm * tc.wn == 8))) {
        if (!prepass_fits) { paella_set_error("gemm: a bf16 LayerNorm-consuming launch on an 8-wave tile needs a workspace (>= 80 MiB + 16 bytes per row) for the row pre-pass"); return PAELLA_ERR_WORKSPACE; }
        const int rc = ln_prepass();
        if (rc != PAELLA_OK) return rc;
    }
    if (bf && !bf16_cfg(cfg)) { paella_set_error("gemm: tile config %d has no bf16-operand variant (10, 18, 19, 30..37 do)", cfg); return PAELLA_ERR_ARG; }
    if ((g.grn_gx || g.ep.grn_gx_out) && (!tc.ring || tc.wm * tc.wn == 8)) { paella_set_error("gemm: the in-epilogue / on-load GRN statistics need a ring tile (got tile %d)", cfg); return PAELLA_ERR_STATE; }
    if (g.grn_gx && (!g.grn_gamma || !g.a_shift || !g.grn_part || g.grn_np <= 0 || g.a_scale || g.ln_stats || g.a_rows_per_sample % 16)) {
        paella_set_error("gemm: bad GRN-from-statistics operand description"); return PAELLA_ERR_ARG;
    }
    if (g.ep.grn_gx_out && (!g.ep.grn_part_out || !(g.ep.grn_rps == 16 || g.ep.grn_rps == BM) || g.M % g.ep.grn_rps ||
                            g.ep.grn_np != ((g.N + BN - 1) / BN) * tc.wn)) {
        paella_set_error("gemm: in-epilogue GRN needs tiles that cover whole samples (rows per sample 16 or %d, got %d) and grn_np = tiles_n * %d", BM, g.ep.grn_rps, tc.wn);
        return PAELLA_ERR_ARG;
    }
    if (tc.ring && !ring_ok(g, BM)) { paella_set_error("gemm: tile config %d (LDS-DMA ring) needs K %% 32 == 0, no implicit convolution and <= 8 samples per tile", cfg); return PAELLA_ERR_ARG; }
    SkPlan p;
    p.tiles_m = (g.M + BM - 1) / BM;
    p.tiles_n = (g.N + BN - 1) / BN;
    const int bk_elems = bf ? 2 * tc.bk : tc.bk;  // a K step is tc.bk * 4 bytes per row
    p.KT = (g.K + bk_elems - 1) / bk_elems;
    const unsigned long long T = (unsigned long long)p.tiles_m * p.tiles_n;
    const unsigned long long U = T * (unsigned long long)p.KT;
    if (U >= (1ull << 31)) { paella_set_error("gemm: problem too large (%llu work units)", U); return PAELLA_ERR_ARG; }
    p.U = (unsigned)U;
    if (G < 1) G = 1;
    if (G > p.U) G = p.U;
    if (g.ln_stats && p.tiles_m > 1) {
        // the LayerNorm-on-load variant computes its row statistics once per workgroup: keep every range inside one tile
        unsigned S = G / (unsigned)T;
        if (S < 1) S = 1;
        while (p.KT % (int)S) --S;
        G = (unsigned)T * S;
    }
    unsigned slab_bytes = 0;
    if (G != (unsigned)T) {  // ranges are not tile-aligned: partial tiles go through slabs + tickets
        const unsigned long long need = 2ull * G * BM * BN * sizeof(float);
        if (!have_ws || need > slab_cap || need >= (1ull << 31) || T > kGemmMaxTickets) {
            paella_set_error("gemm: split-K workspace too small (%llu slab bytes for %u workgroups, %llu tiles)", need, G, T);
            return PAELLA_ERR_WORKSPACE;
        }
        slab_bytes = (unsigned)need;
    }
    p.q = p.U / G;
    p.r = p.U % G;
    p.G = G;
    p.dKT = fast_div_of((unsigned)p.KT); p.dTM = fast_div_of((unsigned)p.tiles_m); p.dQ = fast_div_of(p.q); p.dQ1 = fast_div_of(p.q + 1);
    // grouped rasterisation for launches with many tile rows and columns (test hook: paella_test_gemm_raster)
    // (32-row tiles and skinny problems keep the plain order: their traffic is the weight panel, which m-fastest tiles share best --
    // measured +2 % per image at batch 1 with groups there)
    p.stagger = g_gemm_big_stagger.load(std::memory_order_relaxed);
    const int raster_gm = g_gemm_raster_gm;
    p.gm = (raster_gm > 0 && BM >= 64 && p.tiles_m >= 4 * raster_gm && p.tiles_n >= 4) ? raster_gm : p.tiles_m;
    unsigned* tickets = have_ws ? reinterpret_cast<unsigned*>(ws) : nullptr;
    float* slabs = have_ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + kGemmTicketBytes) : nullptr;
#define GEMM_CASE(id, WMv, WNv, TMv, TNv, PDv) \
    case id: launch_one<WMv, WNv, TMv, TNv, PDv, 32>(g, p, G, slabs, tickets, slab_bytes, st); break;
#define GEMM_CASE64(id, WMv, WNv, TMv, TNv, PDv) \
    case id: launch_one<WMv, WNv, TMv, TNv, PDv, 64>(g, p, G, slabs, tickets, slab_bytes, st); break;
    if (bf) {
        (void)launch_bf_cfg(cfg, g, p, G, slabs, tickets, slab_bytes, st);
        LAUNCH_CHECK_RET();
        return PAELLA_OK;
    }
    switch (cfg) {
        GEMM_CASE(0, 2, 2, 4, 4, 1)
        GEMM_CASE(1, 2, 2, 4, 2, 2)
        GEMM_CASE(2, 2, 2, 2, 2, 2)
        GEMM_CASE(3, 2, 2, 2, 1, 2)
        GEMM_CASE(4, 2, 2, 1, 2, 2)
        GEMM_CASE(5, 2, 2, 1, 1, 2)
        GEMM_CASE(6, 1, 4, 1, 1, 2)
        GEMM_CASE(7, 1, 4, 1, 2, 2)
        GEMM_CASE(8, 1, 4, 2, 2, 2)
        GEMM_CASE(9, 4, 2, 2, 4, 1)
        GEMM_CASE(10, 2, 4, 4, 2, 1)
        GEMM_CASE(11, 4, 1, 2, 2, 2)
        GEMM_CASE(12, 4, 1, 2, 4, 2)
        GEMM_CASE(13, 4, 1, 1, 4, 2)
        GEMM_CASE(14, 4, 2, 2, 2, 2)
        GEMM_CASE(15, 8, 1, 2, 2, 2)
        GEMM_CASE(16, 8, 1, 2, 4, 1)
        GEMM_CASE(17, 8, 1, 1, 4, 2)
        GEMM_CASE(18, 2, 2, 2, 2, 1)
        GEMM_CASE(19, 2, 2, 1, 1, 1)
        GEMM_CASE(20, 4, 1, 2, 4, 1)
        GEMM_CASE(21, 4, 1, 1, 2, 2)
        GEMM_CASE(22, 2, 2, 1, 4, 2)
        GEMM_CASE(23, 1, 4, 2, 1, 2)
        GEMM_CASE64(24, 2, 2, 1, 1, 2)
        GEMM_CASE64(25, 2, 2, 1, 1, 1)
        GEMM_CASE64(26, 2, 2, 2, 2, 1)
        GEMM_CASE64(27, 4, 1, 2, 2, 1)
        GEMM_CASE64(28, 2, 2, 1, 2, 1)
        GEMM_CASE64(29, 4, 2, 2, 2, 1)
        case 30: launch_ring<1, 1, 3>(g, p, G, slabs, tickets, slab_bytes, st); break;
        case 31: launch_ring<1, 1, 4>(g, p, G, slabs, tickets, slab_bytes, st); break;
        case 32: launch_ring<1, 2, 3>(g, p, G, slabs, tickets, slab_bytes, st); break;
        case 33: launch_ring<2, 1, 3>(g, p, G, slabs, tickets, slab_bytes, st); break;
        case 34: launch_ring<2, 2, 3>(g, p, G, slabs, tickets, slab_bytes, st); break;
        case 35: launch_ring<1, 2, 4>(g, p, G, slabs, tickets, slab_bytes, st); break;
        case 36: paella_set_error("gemm: tile config 36 (256x128) exists for bf16 operands only (in fp32 it measured 4-6 %% behind the 64x64 tile: profiles/r04_gemm_big_tile_sweep.txt)"); return PAELLA_ERR_ARG;
        case 37: paella_set_error("gemm: tile config 37 (256x256 ping-pong) exists for bf16 operands only"); return PAELLA_ERR_ARG;
        default: paella_set_error("gemm: bad tile config %d", cfg); return PAELLA_ERR_ARG;
    }
#undef GEMM_CASE
#undef GEMM_CASE64
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

// ---------------------------------------------------------------------------
// head GEMM with the fused sampling tail: one whole tile per workgroup (G = tiles), TAIL instantiations only
// ---------------------------------------------------------------------------
// tile of the fused head + tail: 18 (default) = 64x64 direct-to-LDS, four independent workgroups per CU; 9 = 128x128 (one workgroup per CU: the Philox / log epilogue
// serialises behind the main loop), 14 = 128x64 on 8 waves (two or more workgroups per CU; twice the per-row partials).  A/B through the test hook only
// (profiles/r04_head_tail_and_same_box_ab.txt).
static std::atomic<int> g_tail_tile{18};
extern "C" int paella_test_gemm_tail_tile(int cfg) {
    if (cfg != 9 && cfg != 14 && cfg != 18) { paella_set_error("fused-tail tile must be 9 (128x128), 14 (128x64) or 18 (64x64, direct-to-LDS)"); return PAELLA_ERR_ARG; }
    g_tail_tile = cfg;
    return PAELLA_OK;
}
int gemm_tail_config(int M, int N, bool bf) { return bf ? 18 : (tiles_of_cfg(9, M, N) >= 256 ? g_tail_tile.load() : 2); }  // (bf16 operands: the 64x64 direct-to-LDS tile only)
int gemm_tail_tiles_n(int M, int N, bool bf) {
    const TileCfg& tc = kCfgs[gemm_tail_config(M, N, bf)];
    const int BN = tc.wn * tc.tn * 16;
    return (N + BN - 1) / BN;
}
static int launch_gemm_tail_impl(const GemmArgs& g, hipStream_t st);
int launch_gemm_tail(const GemmArgs& g_in, hipStream_t st) {
    GemmArgs g = g_in;
    g.a_rps_div = fast_div_of(1u);
    g.ep.rps_div = fast_div_of((unsigned)(g.ep.rows_per_sample > 0 ? g.ep.rows_per_sample : 1));
    return prof_bracket(g, st, false, [&]() { return launch_gemm_tail_impl(g, st); });  // (no logits are stored: M*N bytes not counted)
}
static int launch_gemm_tail_impl(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return PAELLA_OK;
    if ((g.K & 3) || (g.N & 3) || (g.lda & 3) || (g.ldw & 3) || g.a_scale || g.ln_stats || !g.ft.part_score || !g.ft.part_idx) {
        paella_set_error("gemm_tail: unsupported arguments (M=%d N=%d K=%d)", g.M, g.N, g.K);
        return PAELLA_ERR_ARG;
    }
    const bool bf = g.A16 != nullptr || g.W16 != nullptr;
    if (bf && (!g.A16 || !g.W16 || !gemm_bf16_ok(g.K, g.lda, g.ldw))) { paella_set_error("gemm_tail: bf16 operands need A16 and W16, K %% 64 == 0, lda / ldw %% 8 == 0"); return PAELLA_ERR_ARG; }
    const int cfg = gemm_tail_config(g.M, g.N, bf);
    const TileCfg& tc = kCfgs[cfg];
    const int BM = tc.wm * tc.tm * 16, BN = tc.wn * tc.tn * 16;
    SkPlan p;
    p.tiles_m = (g.M + BM - 1) / BM;
    p.tiles_n = (g.N + BN - 1) / BN;
    p.KT = bf ? g.K / 64 : (g.K + 31) / 32;
    const unsigned long long T = (unsigned long long)p.tiles_m * p.tiles_n;
    if (T * (unsigned long long)p.KT >= (1ull << 31)) { paella_set_error("gemm_tail: problem too large"); return PAELLA_ERR_ARG; }
    p.U = (unsigned)(T * p.KT);
    // One whole tile per workgroup.  Measured and NOT kept (profiles/r04_head_tail_and_same_box_ab.txt): 512 persistent workgroups walking T / 512 tiles each (round 3: neutral),
    // and the same with the second workgroup of every CU started half a tile period late so that its Philox / log epilogue would run under the other one's main
    // loop (round 4: 20.21 vs 20.21 images/s at configs[2], 112.9-113.3 at batch 32 for every variant) -- the launch is not limited by phase alignment.
    const unsigned long long G = T;
    p.stagger = 0;
    p.q = (unsigned)(p.U / G);
    p.r = 0;
    p.G = (unsigned)G;
    p.dKT = fast_div_of((unsigned)p.KT); p.dTM = fast_div_of((unsigned)p.tiles_m); p.dQ = fast_div_of(p.q); p.dQ1 = fast_div_of(p.q + 1);
    // grouped rasterisation as in the unfused launches: with K = c_out = 256 a tile moves 196 KB of operands for 4.2 MFLOP, and in plain m-fastest order no two
    // tiles that run together share an activation panel -- the whole activation matrix crosses the fabric once per column tile (34 GB per launch at configs[2])
    const int raster_gm = g_gemm_raster_gm;
    p.gm = (raster_gm > 0 && BM >= 64 && p.tiles_m >= 4 * raster_gm && p.tiles_n >= 4) ? raster_gm : p.tiles_m;
    if (bf) hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 2, 2, 1, 0, true, 32, true, 0, true>), dim3((unsigned)G), dim3(256), 0, st, g, p, (float*)nullptr, (unsigned*)nullptr, 0u);
    else if (cfg == 9) hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 2, 4, 1, 0, true>), dim3((unsigned)G), dim3(512), 0, st, g, p, (float*)nullptr, (unsigned*)nullptr, 0u);
    else if (cfg == 14) hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 2, 2, 2, 0, true>), dim3((unsigned)G), dim3(512), 0, st, g, p, (float*)nullptr, (unsigned*)nullptr, 0u);
    else if (cfg == 18 && g.K % 32 == 0 && g_gemm_dma) hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 2, 2, 1, 0, true, 32, true>), dim3((unsigned)G), dim3(256), 0, st, g, p, (float*)nullptr, (unsigned*)nullptr, 0u);
    else hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 2, 2, 2, 0, true>), dim3((unsigned)G), dim3(256), 0, st, g, p, (float*)nullptr, (unsigned*)nullptr, 0u);
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}

int launch_gemm(const GemmArgs& g, void* ws, size_t ws_bytes, hipStream_t st) {
    return launch_gemm_cfg(g, -1, 1, ws, ws_bytes, st);
}

// Zero the ticket header of a freshly allocated workspace (every workspace handed to a paella_* entry point starts with
// it; the kernels leave it zero).  Stream-ordered, no host synchronisation.
extern "C" int paella_workspace_init(void* ws, size_t ws_bytes, void* stream) {
    if (!ws || ws_bytes < kGemmTicketBytes) { paella_set_error("workspace smaller than its %zu-byte header", (size_t)kGemmTicketBytes); return PAELLA_ERR_WORKSPACE; }
    HIP_CHECK_RET(hipMemsetAsync(ws, 0, kGemmTicketBytes, (hipStream_t)stream));
    return PAELLA_OK;
}
extern "C" size_t paella_workspace_header_bytes(void) { return kGemmTicketBytes; }
