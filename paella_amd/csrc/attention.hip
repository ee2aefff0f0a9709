// Flash-style fp32 attention for gfx950 over [self keys | conditioning keys]
// (reference src/modules.py:7-19 Attention2D, :65-79 AttnBlock, nn.MultiheadAttention semantics;
//  utils/alter_attention.py:19-36 for the optional post-softmax per-key weights).
//
// One WORKGROUP owns 16 query rows of one (sample, head); its 4 waves split the key tiles (wave w takes tiles w, w+4, ...)
// and merge their online-softmax states (m, l, O) through LDS in fixed wave order -- at batch-1 sampling sizes
// (64 queries x 68 keys) the kernel is a latency chain, so the chain is cut 4x and 4x more CUs are engaged.
// Both contractions run on the exact-fp32
// matrix cores (v_mfma_f32_16x16x4_f32) in "transposed" form so that no cross-lane data movement
// is needed between them:
//   S^T[key][q]  = K . Q^T    -> lane (r16, kq) holds S^T[key = 4*kq + r][q = r16]
//   O^T[d][q]   += V^T . P^T  -> the P values a lane holds are exactly its B-operand for this product
// so the online-softmax statistics of query q live in the lanes with r16 == q for both steps, row
// max / sum are two xor-shuffles (16, 32) and the rescale of O^T is lane-local.
// head_dim must be a multiple of 16 (80 for the released model); key tiles are 16 wide and the
// ragged tail / segment boundary is handled by masking scores, never by conditional loads: every K/V
// load is unconditional from a clamped in-bounds row so the next tile's loads are all in flight while the
// current tile computes (two register sets, software pipelined; batch-1 sampling is latency-bound here).
#include "common.h"
#include "test_hooks.h"
#include <atomic>
#include <math.h>
#include <type_traits>

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));  // bf16 output of the opt-in fast mode (AttnArgs::out16)

// exp on the hardware exp2 (v_exp_f32 of x * log2 e: 2 instructions, relative error ~1e-7 * (1 + |x|)) instead of the libm sequence (14 instructions, 5 calls
// per 16-key step and lane: the softmax VALU phase is where the matrix core idles -- profiles/r05_attention_pmc_and_probe.txt).  exp_fast(-inf) = 0 as the
// masking relies on.  Every attention kernel uses this one function, so the LDS-staged and register-fed kernels stay bit-identical to each other.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// KSPLIT = true : one workgroup per 16 queries, its 4 waves split the key tiles (latency-bound small grids)
// KSPLIT = false: one workgroup per 64 queries, each wave owns 16 queries and walks all key tiles (K/V re-read 16x less)
template <int DT, bool KSPLIT>  // DT = head_dim / 16
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q0 = KSPLIT ? blockIdx.x * 16 : (blockIdx.x * 4 + wave) * 16;
    if (!KSPLIT && q0 >= a.Lq) return;
    const int h = blockIdx.y, b = blockIdx.z;
    const int r16 = lane & 15, kq = lane >> 4;
    constexpr int D = DT * 16;
    const int Lk = a.Lself + a.Lcond;
    const int ntiles = (Lk + 15) / 16;

    // Q fragment: lane supplies Q[q0 + r16][16*j + 4*kq + e]  (rows past Lq are clamped; their outputs are not stored)
    f32x4 qf[DT];
    {
        const int q = min(q0 + r16, a.Lq - 1);
        const float* qp = a.q + ((size_t)b * a.Lq + q) * a.ldq + h * D + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) qf[j] = *reinterpret_cast<const f32x4*>(qp + j * 16);
    }
    const float* ks_base = a.Lself ? a.k_self + (size_t)b * a.Lself * a.ld_self + h * D : nullptr;
    const float* vs_base = a.Lself ? a.v_self + (size_t)b * a.Lself * a.ld_self + h * D : nullptr;
    const float* kc_base = a.Lcond ? a.k_cond + (size_t)b * a.Lcond * a.ld_cond + h * D : nullptr;
    const float* vc_base = a.Lcond ? a.v_cond + (size_t)b * a.Lcond * a.ld_cond + h * D : nullptr;

    auto krow = [&](int key) -> const float* {  // key clamped to [0, Lk)
        key = min(key, Lk - 1);
        return key < a.Lself ? ks_base + (size_t)key * a.ld_self : kc_base + (size_t)(key - a.Lself) * a.ld_cond;
    };
    auto vrow = [&](int key) -> const float* {
        key = min(key, Lk - 1);
        return key < a.Lself ? vs_base + (size_t)key * a.ld_self : vc_base + (size_t)(key - a.Lself) * a.ld_cond;
    };
    // K operand: lane supplies K[kt*16 + r16][16*j + 4*kq + e]; V^T operand: V[kt*16 + 4*kq + e][j*16 + r16]
    auto load_tile = [&](int kt, f32x4 (&kf)[DT], float (&vf)[DT][4]) {
        const float* kp = krow(kt * 16 + r16) + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kp + j * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* vp = vrow(kt * 16 + kq * 4 + e) + r16;
#pragma unroll
            for (int j = 0; j < DT; ++j) vf[j][e] = vp[j * 16];
        }
    };

    f32x4 oacc[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    auto process = [&](int kt, const f32x4 (&kf)[DT], const float (&vf)[DT][4]) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < DT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][e], qf[j][e], s, 0, 0, 0);
        // online softmax for query r16 over keys kt*16 + 4*kq + r
        float p[4];
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + kq * 4 + r;
            p[r] = key < Lk ? s[r] * a.scale : -INFINITY;
            mt = fmaxf(mt, p[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp_fast(m_run - m_new);  // first tile: exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = exp_fast(p[r] - m_new);  // masked keys: exp(-inf) = 0, which also zeroes their (clamped) V rows
            psum += p[r];
        }
        l_run = l_run * alpha + psum;  // per-lane partial; lanes of equal r16 are combined at the end
        m_run = m_new;
        if (a.key_weights) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int wi = kt * 16 + kq * 4 + r - (Lk - a.n_kw);
                const float wv = a.key_weights[min(max(wi, 0), a.n_kw - 1)];
                if (wi >= 0 && wi < a.n_kw) p[r] *= wv;
            }
        }
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            oacc[j] *= alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j][e], p[e], oacc[j], 0, 0, 0);
        }
    };

    f32x4 kfA[DT], kfB[DT];
    float vfA[DT][4], vfB[DT][4];
    // this wave's key tiles: KSPLIT -> wave, wave + 4, ...; else all of them (software pipelined, two register sets)
    constexpr int KSTEP = KSPLIT ? 4 : 1;
    const int kt0 = KSPLIT ? wave : 0;
    load_tile(kt0, kfA, vfA);
    for (int kt = kt0; kt < ntiles; kt += 2 * KSTEP) {
        load_tile(kt + KSTEP, kfB, vfB);
        __builtin_amdgcn_sched_barrier(0);
        process(kt, kfA, vfA);
        if (kt + KSTEP < ntiles) {  // wave-uniform
            load_tile(kt + 2 * KSTEP, kfA, vfA);
            __builtin_amdgcn_sched_barrier(0);
            process(kt + KSTEP, kfB, vfB);
        }
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if constexpr (!KSPLIT) {
        const float inv1 = 1.0f / l;
        const int q1 = q0 + r16;
        if (q1 < a.Lq) {
            const size_t oo = ((size_t)b * a.Lq + q1) * a.ldo + h * D + kq * 4;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                if (a.out16) *reinterpret_cast<bf16x4*>(a.out16 + oo + j * 16) = __builtin_convertvector(oacc[j] * inv1, bf16x4);
                else *reinterpret_cast<f32x4*>(a.out + oo + j * 16) = oacc[j] * inv1;
            }
        }
        return;
    }

    // ---- merge the 4 waves' (m, l, O^T) in fixed order ----
    __shared__ float s_m[4][64], s_l[4][64];
    __shared__ __attribute__((aligned(16))) float s_o[4][DT][64][4];
    s_m[wave][lane] = m_run;
    s_l[wave][lane] = l;
#pragma unroll
    for (int j = 0; j < DT; ++j) *reinterpret_cast<f32x4*>(&s_o[wave][j][lane][0]) = oacc[j];
    __syncthreads();
    float mw[4], m_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = s_m[w][lane]; m_all = fmaxf(m_all, mw[w]); }
    float l_all = 0.f, f[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        f[w] = exp_fast(mw[w] - m_all);  // empty waves: exp(-inf) = 0
        l_all += s_l[w][lane] * f[w];
    }
    const float inv = 1.0f / l_all;
    const int q = q0 + r16;
    // wave w finalises the output d-tiles j = w, w+4, ...
    for (int j = wave; j < DT; j += 4) {
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) o += *reinterpret_cast<const f32x4*>(&s_o[w][j][lane][0]) * f[w];
        if (q < a.Lq) {
            const size_t oo = ((size_t)b * a.Lq + q) * a.ldo + h * D + kq * 4 + j * 16;
            if (a.out16) *reinterpret_cast<bf16x4*>(a.out16 + oo) = __builtin_convertvector(o * inv, bf16x4);
            else *reinterpret_cast<f32x4*>(a.out + oo) = o * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Large query counts (>= 256 queries per sample: 512 px grids and up, BASELINE configs 3-5 where attention is up to ~40 % of a
// block's FLOPs): one workgroup = 64 queries of one (sample, head); its 4 waves own 16 queries each and SHARE the K/V tiles, which
// are fetched from global memory ONCE per workgroup and staged in LDS (32 keys per stage, two stages: tile t+1 lands while tile t multiplies).
// Per 32-key stage a wave issues both sub-tiles' S chains together (two accumulators, alternating: no two consecutive MFMAs share one; the second chain's
// results are in flight while the first sub-tile's softmax runs), then the two online-softmax / PV halves in key order with the PV MFMAs ordered so that
// consecutive ones hit different accumulators.  Same arithmetic in the same order as the register-fed kernel above (16-key steps walked in key order per
// wave): results are bit-identical to it, whatever the staging.
//
// STAGING (template parameter STG).  The round-5 ablations (profiles/r05_attention_staging_probe.txt) put the kernel's idle matrix-core time on the staging, not on
// the softmax: with no softmax at all 0.73 -> 0.77 of peak, with no staging and no barrier in the loop 0.73 -> 0.84 (both: 0.90).
//   0 = through registers (rounds 2-5; kept behind the test hook for A/B): per stage and thread 6 global b128 loads with their clamp / select / divide address
//       arithmetic, 6 ds_write_b128 (13 issue cycles each), 24 staging VGPRs.
//   1 = direct-to-LDS buffer loads: one wave instruction lands 64 consecutive 16-byte chunks, so a tile is [32 rows][D/4 + 1 chunks] (the pad chunk of a row is a
//       duplicate of its last chunk) rounded up to whole wave instructions; a stage whose 32 keys are all self rows or all conditioning rows (every stage but at
//       most two) addresses with per-lane offsets computed ONCE plus one scalar offset per stage; the mixed stages clamp and select per lane.  No staging
//       registers, no ds_write: 138 -> 110 VGPRs at head_dim 80 and 0.70-0.72 -> 0.77-0.80 of the fp32 MFMA peak.
//       K tile: the K.Q^T fragment of a lane is 16 contiguous bytes of one key row -> ds_read_b128; row pitch D/4 + 1 (odd) 16-byte slots: 2-way conflicts remain
//       (a b128 lane group is 8 rows of one chunk column + 8 OTHER rows of the next; tests/test_attention_layout.py mirrors both layouts against the banking rules).
//       V tile: the V^T.P^T fragment of a lane is V[key = 4*kq + e][d = 16*j + r16] -> ds_read_b32; lanes r16 sweep 16 consecutive banks and
//       4*(D + 4) mod 32 = 16 puts kq = 0 / 1 on disjoint halves: conflict-free.
//   2 = STG 1 without any padding (odd D/16 only): a tile is exactly [32][D] floats = 10 wave instructions at head_dim 80 and a workgroup's two stages are
//       40 960 B, so FOUR workgroups fit the CU's 160 KB (4 waves per SIMD instead of 3).  Bank conflicts are avoided by WHERE the DMA puts things, which costs
//       nothing (the source address of a lane is free):  K tile: LDS chunk position p of row r holds source chunk p ^ f((r >> 2) & 3), f = {0, 3, 2, 1} (an XOR on
//       the low two bits stays inside the row's D/4 chunks); every ds_read_b128 lane group then covers 16 distinct 16-byte slots.  V tile: LDS row r holds key
//       r ^ ((r >> 2) & 1), so the rows the kq = 0 / 1 (2 / 3) lanes of a ds_read_b32 group read have opposite parity = disjoint 16-bank halves.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
#endif
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t attn_rsrc(const float* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
}
template <int DT, int STG>
__global__ __launch_bounds__(256) void attention_lds_kernel(AttnArgs args) {
    constexpr int D = DT * 16, D4 = D / 4;
    constexpr int KTILE = 32;                          // keys per LDS stage
    constexpr int PITCH = STG == 2 ? D : D + 4;        // floats per LDS row
    constexpr int RCH = PITCH / 4;                     // 16-byte chunks per LDS row
    constexpr int NI = (KTILE * RCH + 63) / 64;        // direct-to-LDS: wave instructions per tile
    constexpr int NPW = (NI + 3) / 4;                  // ... per wave and tensor
    constexpr int VT_OFF = STG ? NI * 256 : KTILE * PITCH;  // V tile offset inside a stage (floats)
    constexpr int STAGE = 2 * VT_OFF;                  // K tile + V tile
    constexpr int NITEM = KTILE * D4;                  // register staging: float4 items per tensor per stage
    constexpr int NL = STG ? 1 : (NITEM + 255) / 256;
    static_assert(STG != 2 || (DT & 1), "the unpadded layout's swizzle needs an odd number of 64-byte blocks per row");
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    // scalar copies of the argument fields: lambdas that capture the argument STRUCT by reference make hipcc spill it to scratch
    const int Lq = args.Lq, Lself = args.Lself, Lcond = args.Lcond, ld_self = args.ld_self, ld_cond = args.ld_cond, ldq = args.ldq, ldo = args.ldo;
    const int n_kw = args.n_kw;
    const float scale = args.scale;
    const float* const key_weights = args.key_weights;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    const int h = blockIdx.y, b = blockIdx.z;
    const int r16 = lane & 15, kq = lane >> 4;
    const int Lk = Lself + Lcond;
    const int ntiles = (Lk + KTILE - 1) / KTILE;

    f32x4 qf[DT];
    {
        const int q = min(q0 + r16, Lq - 1);  // rows past Lq are clamped; their outputs are not stored
        const float* qp = args.q + ((size_t)b * Lq + q) * ldq + h * D + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) qf[j] = *reinterpret_cast<const f32x4*>(qp + j * 16);
    }
    const float* ks_base = Lself ? args.k_self + (size_t)b * Lself * ld_self + h * D : nullptr;
    const float* vs_base = Lself ? args.v_self + (size_t)b * Lself * ld_self + h * D : nullptr;
    const float* kc_base = Lcond ? args.k_cond + (size_t)b * Lcond * ld_cond + h * D : nullptr;
    const float* vc_base = Lcond ? args.v_cond + (size_t)b * Lcond * ld_cond + h * D : nullptr;

    // ---- STG 0: item = (key, c4) of the K tile and the same item of the V tile; loads are unconditional from clamped rows (masked keys get score -inf below).
    // K and V go through separate, fully unrolled loops: a run-time K/V selector makes hipcc build a pointer table in scratch.
    f32x4 stk[NL], stv[NL];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = min(tid + i * 256, NITEM - 1);
            const int key_in = idx / D4, c4 = idx - key_in * D4;
            const int key = min(kt * KTILE + key_in, Lk - 1);
            const bool self = key < Lself;
            const size_t off = self ? (size_t)key * ld_self : (size_t)(key - Lself) * ld_cond;
            stk[i] = *reinterpret_cast<const f32x4*>((self ? ks_base : kc_base) + off + c4 * 4);
            stv[i] = *reinterpret_cast<const f32x4*>((self ? vs_base : vc_base) + off + c4 * 4);
        }
    };
    auto store_tile = [&](int slot) __attribute__((always_inline)) {
        float* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * 256;
            if (NL * 256 == NITEM || idx < NITEM) {
                const int key_in = idx / D4, c4 = idx - key_in * D4;
                *reinterpret_cast<f32x4*>(base + key_in * PITCH + c4 * 4) = stk[i];
                *reinterpret_cast<f32x4*>(base + VT_OFF + key_in * PITCH + c4 * 4) = stv[i];
            }
        }
    };
    // ---- STG 1 / 2: wave w issues instructions m = 4 i + w (i < NPW, m < NI) of the K tile and the same of the V tile; lane l of instruction m lands LDS chunk
    // g = 64 m + l = (row g / RCH, position g % RCH) of the tile (rows past 31: the tile's rounding, a re-load of row 31 into unused LDS).
    // krow / vrow: the key row (0..31) the chunk comes from; kcb / vcb: its byte offset inside that row.
    int krow[NPW], vrow[NPW];
    unsigned kcb[NPW], vcb[NPW];
    __amdgpu_buffer_rsrc_t rs_ks, rs_vs, rs_kc, rs_vc;
    if constexpr (STG != 0) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int g = (4 * i + wave) * 64 + lane;
            const int row = min(g / RCH, KTILE - 1), pos = min(g - (g / RCH) * RCH, D4 - 1);
            if constexpr (STG == 2) {
                krow[i] = row;
                kcb[i] = (unsigned)((pos ^ ((4 - ((row >> 2) & 3)) & 3)) * 16);  // f = {0, 3, 2, 1}
                vrow[i] = row ^ ((row >> 2) & 1);
                vcb[i] = (unsigned)(pos * 16);
            } else {
                krow[i] = vrow[i] = row;
                kcb[i] = vcb[i] = (unsigned)(pos * 16);
            }
        }
        const size_t self_bytes = (size_t)max(Lself, 1) * ld_self * 4, cond_bytes = (size_t)max(Lcond, 1) * ld_cond * 4;
        rs_ks = attn_rsrc(Lself ? ks_base : kc_base, self_bytes);
        rs_vs = attn_rsrc(Lself ? vs_base : vc_base, self_bytes);
        rs_kc = attn_rsrc(Lcond ? kc_base : ks_base, cond_bytes);
        rs_vc = attn_rsrc(Lcond ? vc_base : vs_base, cond_bytes);
    }
    auto dma_stage = [&](int kt, int slot) __attribute__((always_inline)) {
        float* base = smem + slot * STAGE;
        const int kb = kt * KTILE;
        if (kb + KTILE <= Lself) {  // 32 self rows
            const int so = kb * ld_self * 4;
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                if (4 * i + wave < NI) {
                    attn_dma16(rs_ks, base + (4 * i + wave) * 256, (unsigned)(krow[i] * ld_self * 4) + kcb[i], so);
                    attn_dma16(rs_vs, base + VT_OFF + (4 * i + wave) * 256, (unsigned)(vrow[i] * ld_self * 4) + vcb[i], so);
                }
        } else if (kb >= Lself && kb + KTILE <= Lk) {  // 32 conditioning rows
            const int so = (kb - Lself) * ld_cond * 4;
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                if (4 * i + wave < NI) {
                    attn_dma16(rs_kc, base + (4 * i + wave) * 256, (unsigned)(krow[i] * ld_cond * 4) + kcb[i], so);
                    attn_dma16(rs_vc, base + VT_OFF + (4 * i + wave) * 256, (unsigned)(vrow[i] * ld_cond * 4) + vcb[i], so);
                }
        } else {  // the self / conditioning boundary or the ragged last stage: clamp and select per lane (masked keys get score -inf below)
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                if (4 * i + wave < NI) {
                    float* dk = base + (4 * i + wave) * 256;
                    const int kk = min(kb + krow[i], Lk - 1), kv = min(kb + vrow[i], Lk - 1);
                    if (kk < Lself) attn_dma16(rs_ks, dk, (unsigned)(kk * ld_self * 4) + kcb[i], 0);
                    else attn_dma16(rs_kc, dk, (unsigned)((kk - Lself) * ld_cond * 4) + kcb[i], 0);
                    if (kv < Lself) attn_dma16(rs_vs, dk + VT_OFF, (unsigned)(kv * ld_self * 4) + vcb[i], 0);
                    else attn_dma16(rs_vc, dk + VT_OFF, (unsigned)((kv - Lself) * ld_cond * 4) + vcb[i], 0);
                }
        }
    };
    // fragment addresses inside a tile (floats): K row r16 (+16), this lane's 16-byte chunk kq (+4 j); V row 4 kq + e (+16), column r16 (+16 j)
    const int kfrag = STG == 2 ? r16 * PITCH + ((kq ^ ((4 - ((r16 >> 2) & 3)) & 3)) * 4) : r16 * PITCH + kq * 4;
    auto vfrag = [&](int e) __attribute__((always_inline)) { return (STG == 2 ? ((kq * 4 + e) ^ (kq & 1)) : kq * 4 + e) * PITCH + r16; };

    f32x4 oacc[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // second half of a 16-key step: online softmax over S^T (lane (r16, kq) holds the scores of query r16 against keys key0 + 4 * kq + 0..3), then O^T += V^T . P^T
    auto softmax_pv = [&](int key0, const f32x4 s, const float* Vs) __attribute__((always_inline)) {
        float vf[DT][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* vp = Vs + vfrag(e);
#pragma unroll
            for (int j = 0; j < DT; ++j) vf[j][e] = vp[j * 16];
        }
        float p[4];
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + kq * 4 + r;
            p[r] = key < Lk ? s[r] * scale : -INFINITY;
            mt = fmaxf(mt, p[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp_fast(m_run - m_new);  // first tile: exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = exp_fast(p[r] - m_new);  // masked keys: exp(-inf) = 0, which also zeroes their (clamped) V rows
            psum += p[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (key_weights) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int wi = key0 + kq * 4 + r - (Lk - n_kw);
                const float wv = key_weights[min(max(wi, 0), n_kw - 1)];
                if (wi >= 0 && wi < n_kw) p[r] *= wv;
            }
        }
#pragma unroll
        for (int j = 0; j < DT; ++j) oacc[j] *= alpha;
        // the four accumulations per oacc[j] in the order e = 0..3; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < DT; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j][e], p[e], oacc[j], 0, 0, 0);
    };
    // a whole 32-key stage: both S chains first (two accumulators, alternating), then the two softmax / PV halves in key order
    auto process32 = [&](int key0, const float* Ks, const float* Vs, bool second) __attribute__((always_inline)) {
        f32x4 kf0[DT], kf1[DT];
        const float* kp = Ks + kfrag;
#pragma unroll
        for (int j = 0; j < DT; ++j) { kf0[j] = *reinterpret_cast<const f32x4*>(kp + j * 16); kf1[j] = *reinterpret_cast<const f32x4*>(kp + 16 * PITCH + j * 16); }
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int j = 0; j < DT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf0[j][e], qf[j][e], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf1[j][e], qf[j][e], s1, 0, 0, 0);
            }
        softmax_pv(key0, s0, Vs);
        if (second) softmax_pv(key0 + 16, s1, Vs + 16 * PITCH);
    };

    if constexpr (STG != 0) {
        dma_stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int slot = kt & 1;
        if constexpr (STG != 0) {
            if (kt + 1 < ntiles) dma_stage(kt + 1, slot ^ 1);  // lands while this stage computes; slot ^ 1 was released by the barrier that ended stage kt - 1
        } else {
            load_tile(min(kt + 1, ntiles - 1));  // the last iteration re-reads its own tile (L1/L2 hit, never consumed)
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* Ks = smem + slot * STAGE;
        process32(kt * KTILE, Ks, Ks + VT_OFF, kt * KTILE + 16 < Lk);  // (a fully masked second half is multiplied -- clamped rows -- and dropped)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STG != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the next stage has landed before the barrier that publishes it
        else store_tile(slot ^ 1);
        __syncthreads();
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + r16;
    if (q < Lq) {
        const size_t oo = ((size_t)b * Lq + q) * ldo + h * D + kq * 4;
        unsigned short* const o16 = args.out16;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            if (o16) *reinterpret_cast<bf16x4*>(o16 + oo + j * 16) = __builtin_convertvector(oacc[j] * inv, bf16x4);
            else *reinterpret_cast<f32x4*>(args.out + oo + j * 16) = oacc[j] * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// OPT-IN bf16 fast mode (outside the fp32 parity contract), >= 256 queries: the LDS-staged kernel with both contractions on v_mfma_f32_16x16x16_bf16.
// q / self K / self V arrive as bf16 (the in-projection's epilogue copy), the conditioning K / V as the fp32 cache, rounded while staged; the online softmax
// stays fp32 (exp_fast), the probabilities are rounded to bf16 for the second contraction (they are its B operand as they sit in the lane), output bf16.
//   K tile  [32 keys][D + 8] bf16: a lane's S fragment is 8 contiguous bytes of one key row (ds_read_b64); pitch 2D + 16 bytes = 16 * odd -> conflict-free
//   V tile  TRANSPOSED [D][32 + 8] bf16: a lane's PV fragment is V[key = 4 kq .. + 3][d = 16 j + r16] = 8 contiguous bytes of row d (ds_read_b64)
// The matrix cores are no longer the bound (40 + 40 MFMA cycles per 16-key step against 1 280 in fp32): the step is its softmax VALU and LDS traffic.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <int DT>
__global__ __launch_bounds__(256) void attention_bf16_kernel(AttnArgs args) {
    constexpr int D = DT * 16, D8 = D / 8;
    constexpr int KTILE = 32;
    constexpr int PK = D + 8;                  // K row pitch (bf16 elements)
    constexpr int PV = KTILE + 8;              // V^T row pitch
    constexpr int STAGE = KTILE * PK + D * PV; // elements per stage
    constexpr int NITEM = KTILE * D8;          // 16-byte items per tensor and stage
    constexpr int NL = (NITEM + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];

    const int Lq = args.Lq, Lself = args.Lself, Lcond = args.Lcond, ld16 = args.ld16, ld_cond = args.ld_cond, ldo = args.ldo;
    const int n_kw = args.n_kw;
    const float scale = args.scale;
    const float* const key_weights = args.key_weights;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    const int h = blockIdx.y, b = blockIdx.z;
    const int r16 = lane & 15, kq = lane >> 4;
    const int Lk = Lself + Lcond;
    const int ntiles = (Lk + KTILE - 1) / KTILE;

    s16x4 qf[DT];  // Q[q0 + r16][16 j + 4 kq .. + 3]
    {
        const int q = min(q0 + r16, Lq - 1);
        const unsigned short* qp = args.q16 + ((size_t)b * Lq + q) * ld16 + h * D + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) qf[j] = *reinterpret_cast<const s16x4*>(qp + j * 16);
    }
    const unsigned short* ks_base = Lself ? args.k_self16 + (size_t)b * Lself * ld16 + h * D : nullptr;
    const unsigned short* vs_base = Lself ? args.v_self16 + (size_t)b * Lself * ld16 + h * D : nullptr;
    const float* kc_base = Lcond ? args.k_cond + (size_t)b * Lcond * ld_cond + h * D : nullptr;
    const float* vc_base = Lcond ? args.v_cond + (size_t)b * Lcond * ld_cond + h * D : nullptr;

    bf16x8 stk[NL], stv[NL];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = min(tid + i * 256, NITEM - 1);
            const int key_in = idx / D8, c8 = idx - key_in * D8;
            const int key = min(kt * KTILE + key_in, Lk - 1);
            if (key < Lself) {
                const size_t off = (size_t)key * ld16 + c8 * 8;
                stk[i] = *reinterpret_cast<const bf16x8*>(ks_base + off);
                stv[i] = *reinterpret_cast<const bf16x8*>(vs_base + off);
            } else {
                const size_t off = (size_t)(key - Lself) * ld_cond + c8 * 8;
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(kc_base + off), k1 = *reinterpret_cast<const f32x4*>(kc_base + off + 4);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(vc_base + off), v1 = *reinterpret_cast<const f32x4*>(vc_base + off + 4);
                stk[i] = __builtin_convertvector((f32x8){k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]}, bf16x8);
                stv[i] = __builtin_convertvector((f32x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}, bf16x8);
            }
        }
    };
    auto store_tile = [&](int slot) __attribute__((always_inline)) {
        unsigned short* Kt = smem + slot * STAGE;
        unsigned short* Vt = Kt + KTILE * PK;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * 256;
            if (NL * 256 == NITEM || idx < NITEM) {
                const int key_in = idx / D8, c8 = idx - key_in * D8;
                *reinterpret_cast<bf16x8*>(Kt + key_in * PK + c8 * 8) = stk[i];
                const __attribute__((ext_vector_type(8))) unsigned short vb = __builtin_bit_cast(__attribute__((ext_vector_type(8))) unsigned short, stv[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) Vt[(c8 * 8 + e) * PV + key_in] = vb[e];  // transposed: row = d, column = key
            }
        }
    };

    f32x4 oacc[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    auto process16 = [&](int key0, const unsigned short* Ks, const unsigned short* Vts) __attribute__((always_inline)) {  // Ks: this sub-tile's 16 key rows; Vts: V^T + its 16 key columns
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned short* kp = Ks + r16 * PK + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) s = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const s16x4*>(kp + j * 16), qf[j], s, 0, 0, 0);
        s16x4 vf[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) vf[j] = *reinterpret_cast<const s16x4*>(Vts + (j * 16 + r16) * PV + kq * 4);
        float p[4];
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + kq * 4 + r;
            p[r] = key < Lk ? s[r] * scale : -INFINITY;
            mt = fmaxf(mt, p[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp_fast(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = exp_fast(p[r] - m_new);
            psum += p[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (key_weights) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int wi = key0 + kq * 4 + r - (Lk - n_kw);
                const float wv = key_weights[min(max(wi, 0), n_kw - 1)];
                if (wi >= 0 && wi < n_kw) p[r] *= wv;
            }
        }
        const s16x4 pb = __builtin_bit_cast(s16x4, __builtin_convertvector((f32x4){p[0], p[1], p[2], p[3]}, bf16x4));
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            oacc[j] *= alpha;
            oacc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[j], pb, oacc[j], 0, 0, 0);
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int slot = kt & 1;
        load_tile(min(kt + 1, ntiles - 1));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned short* Ks = smem + slot * STAGE;
        const unsigned short* Vt = Ks + KTILE * PK;
#pragma unroll
        for (int sub = 0; sub < KTILE / 16; ++sub)
            if (kt * KTILE + sub * 16 < Lk) process16(kt * KTILE + sub * 16, Ks + sub * 16 * PK, Vt + sub * 16);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(slot ^ 1);
        __syncthreads();
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + r16;
    if (q < Lq) {
        unsigned short* op = args.out16 + ((size_t)b * Lq + q) * ldo + h * D + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) *reinterpret_cast<bf16x4*>(op + j * 16) = __builtin_convertvector(oacc[j] * inv, bf16x4);
    }
}

static std::atomic<int> g_attn_variant{0};  // test hook (test_hooks.h): 1 = register-fed kernel for large query counts too; 10 / 11 = LDS kernel with staging 0 / 1 (A/B probes; default: staging 2 at odd head_dim / 16, else 1)
extern "C" int paella_test_attention_variant(int v) { g_attn_variant = v; return PAELLA_OK; }

int launch_attention(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.Lq <= 0) return PAELLA_OK;
    if (a.D % 16 || a.D > 128 || a.D <= 0) {
        paella_set_error("attention: head_dim %d unsupported (need a multiple of 16, <= 128)", a.D);
        return PAELLA_ERR_ARG;
    }
    if (a.Lself + a.Lcond <= 0) { paella_set_error("attention: no keys"); return PAELLA_ERR_ARG; }
    if ((a.ldq & 3) || (a.ldo & 3) || (a.Lself && (a.ld_self & 3)) || (a.Lcond && (a.ld_cond & 3))) {
        paella_set_error("attention: leading dimensions must be multiples of 4");
        return PAELLA_ERR_ARG;
    }
    if (a.key_weights && (a.n_kw > a.Lself + a.Lcond || a.n_kw < 1)) { paella_set_error("attention: attn_weights longer than the key sequence"); return PAELLA_ERR_ARG; }
    if (a.q16) {  // opt-in bf16 fast mode (the model only asks for it at >= 256 queries)
        if (!a.out16 || (a.Lself && (!a.k_self16 || !a.v_self16)) || (a.ld16 & 7) || (a.Lcond && (a.ld_cond & 3)) || (a.ldo & 3) || a.D % 16) {
            paella_set_error("attention (bf16): needs out16, bf16 self K / V, ld16 %% 8 == 0");
            return PAELLA_ERR_ARG;
        }
        dim3 grid16((a.Lq + 63) / 64, a.nhead, a.B);
#define ATT16_CASE(n) case n: hipLaunchKernelGGL((attention_bf16_kernel<n>), grid16, dim3(256), 0, st, a); break;
        switch (a.D / 16) { ATT16_CASE(2) ATT16_CASE(3) ATT16_CASE(4) ATT16_CASE(5) ATT16_CASE(6) ATT16_CASE(7) ATT16_CASE(8)
            default: paella_set_error("attention (bf16): head_dim %d unsupported", a.D); return PAELLA_ERR_ARG; }
#undef ATT16_CASE
        LAUNCH_CHECK_RET();
        return PAELLA_OK;
    }
    // small query counts (batch-1 sampling grids) are latency chains -> split keys over waves; large ones re-read K/V
    // once per workgroup, so give a workgroup 64 queries instead
    // (the key split buys CUs and cuts the latency chain 4x when the grid is small; once the 64-query form fills the chip by itself -- >= 1024 workgroups: batch >= 32 with
    // guidance at the 64-position level -- it wins: no cross-wave merge, balanced key tiles.  variant 20 / 21 / 22 force the key-split / the LDS-staged / the register-fed 64-query form at 64..255 queries, A/B)
    const int variant = g_attn_variant.load();
    const long wg64 = (long)((a.Lq + 63) / 64) * a.nhead * a.B;
    const bool big = a.Lq >= 64 && (variant == 21 || variant == 22 || (variant != 20 && wg64 >= 1024));  // (22: the register-fed 64-query kernel instead of the LDS-staged one)
    const bool ksplit = a.Lq < 256 && !big;
    // the direct-to-LDS stagings address K / V with 32-bit byte offsets behind a buffer descriptor whose num_records is clamped to 2 GiB: a per-sample K / V
    // span beyond that would be range-checked to zeros silently -- such a call takes the register-fed kernel, which uses 64-bit pointers
    const bool span32 = (size_t)a.Lself * (size_t)a.ld_self * 4 < ((size_t)1 << 31) && (size_t)a.Lcond * (size_t)a.ld_cond * 4 < ((size_t)1 << 31);
    const bool lds = !ksplit && variant != 1 && variant != 22 && span32;
    dim3 grid(ksplit ? (a.Lq + 15) / 16 : (a.Lq + 63) / 64, a.nhead, a.B);
#define ATT_CASE(n)                                                                                                     \
    case n:                                                                                                             \
        if (ksplit) hipLaunchKernelGGL((attention_kernel<n, true>), grid, dim3(256), 0, st, a);                         \
        else if (lds && variant == 10) hipLaunchKernelGGL((attention_lds_kernel<n, 0>), grid, dim3(256), 0, st, a);     \
        else if (lds && variant == 11) hipLaunchKernelGGL((attention_lds_kernel<n, 1>), grid, dim3(256), 0, st, a);     \
        else if (lds) hipLaunchKernelGGL((attention_lds_kernel<n, (n & 1) ? 2 : 1>), grid, dim3(256), 0, st, a);        \
        else hipLaunchKernelGGL((attention_kernel<n, false>), grid, dim3(256), 0, st, a);                               \
        break;
    switch (a.D / 16) {
        case 1:  // head_dim 16 (toy models): the LDS-staged instantiation spills; the register-fed kernel serves large query counts there
            if (ksplit) hipLaunchKernelGGL((attention_kernel<1, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((attention_kernel<1, false>), grid, dim3(256), 0, st, a);
            break;
        ATT_CASE(2) ATT_CASE(3) ATT_CASE(4) ATT_CASE(5) ATT_CASE(6) ATT_CASE(7) ATT_CASE(8)
    }
#undef ATT_CASE
    switch (0) {
        default: break;
    }
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
