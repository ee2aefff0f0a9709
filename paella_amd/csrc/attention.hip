// Flash-style fp32 attention for gfx950 over [self keys | conditioning keys]
// (reference src/modules.py:7-19 Attention2D, :65-79 AttnBlock, nn.MultiheadAttention semantics;
//  utils/alter_attention.py:19-36 for the optional post-softmax per-key weights).
//
// One wave owns 16 query rows of one (sample, head).  Both contractions run on the exact-fp32
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
#include <math.h>

template <int DT>  // DT = head_dim / 16
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= a.Lq) return;
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
        const float alpha = expf(m_run - m_new);  // first tile: exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = expf(p[r] - m_new);  // masked keys: exp(-inf) = 0, which also zeroes their (clamped) V rows
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
    load_tile(0, kfA, vfA);
    for (int kt = 0; kt < ntiles; kt += 2) {
        load_tile(kt + 1, kfB, vfB);
        __builtin_amdgcn_sched_barrier(0);
        process(kt, kfA, vfA);
        if (kt + 1 < ntiles) {  // wave-uniform
            load_tile(kt + 2, kfA, vfA);
            __builtin_amdgcn_sched_barrier(0);
            process(kt + 1, kfB, vfB);
        }
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + r16;
    if (q < a.Lq) {
        float* op = a.out + ((size_t)b * a.Lq + q) * a.ldo + h * D + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) *reinterpret_cast<f32x4*>(op + j * 16) = oacc[j] * inv;
    }
}

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
    dim3 grid((a.Lq + 63) / 64, a.nhead, a.B);
    switch (a.D / 16) {
        case 1: hipLaunchKernelGGL((attention_kernel<1>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((attention_kernel<2>), grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((attention_kernel<3>), grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((attention_kernel<4>), grid, dim3(256), 0, st, a); break;
        case 5: hipLaunchKernelGGL((attention_kernel<5>), grid, dim3(256), 0, st, a); break;
        case 6: hipLaunchKernelGGL((attention_kernel<6>), grid, dim3(256), 0, st, a); break;
        case 7: hipLaunchKernelGGL((attention_kernel<7>), grid, dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL((attention_kernel<8>), grid, dim3(256), 0, st, a); break;
    }
    LAUNCH_CHECK_RET();
    return PAELLA_OK;
}
