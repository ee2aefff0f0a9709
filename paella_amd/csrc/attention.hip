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
// ragged tail / segment boundary is handled by masking.
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

    // Q fragment: lane supplies Q[q0 + r16][16*j + 4*kq + e]
    f32x4 qf[DT];
    {
        const int q = q0 + r16;
        const float* qp = a.q + ((size_t)b * a.Lq + q) * a.ldq + h * D + kq * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) qf[j] = (q < a.Lq) ? *reinterpret_cast<const f32x4*>(qp + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f32x4 oacc[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (Lk + 15) / 16;
    for (int kt = 0; kt < ntiles; ++kt) {
        // ---- S^T tile = K_tile . Q^T ----
        const int keyA = kt * 16 + r16;  // key row this lane feeds as the A operand
        const float* kp = nullptr;
        if (keyA < a.Lself) kp = a.k_self + ((size_t)b * a.Lself + keyA) * a.ld_self + h * D + kq * 4;
        else if (keyA < Lk) kp = a.k_cond + ((size_t)b * a.Lcond + (keyA - a.Lself)) * a.ld_cond + h * D + kq * 4;
        f32x4 kf[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) kf[j] = kp ? *reinterpret_cast<const f32x4*>(kp + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};

        // V^T operand loads are independent of the softmax: issue them early.
        // lane supplies V[key = kt*16 + 4*kq + e][dt*16 + r16]
        float vf[DT][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = kt * 16 + kq * 4 + e;
            const float* vp = nullptr;
            if (key < a.Lself) vp = a.v_self + ((size_t)b * a.Lself + key) * a.ld_self + h * D + r16;
            else if (key < Lk) vp = a.v_cond + ((size_t)b * a.Lcond + (key - a.Lself)) * a.ld_cond + h * D + r16;
#pragma unroll
            for (int j = 0; j < DT; ++j) vf[j][e] = vp ? vp[j * 16] : 0.f;
        }

        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < DT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][e], qf[j][e], s, 0, 0, 0);

        // ---- online softmax for query r16 over keys 4*kq + r (r = 0..3) ----
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
            p[r] = expf(p[r] - m_new);
            psum += p[r];
        }
        l_run = l_run * alpha + psum;  // per-lane partial; lanes of equal r16 are combined at the end
        m_run = m_new;
        if (a.key_weights) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + kq * 4 + r;
                const int wi = key - (Lk - a.n_kw);
                if (wi >= 0 && key < Lk) p[r] *= a.key_weights[wi];
            }
        }
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            oacc[j] *= alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j][e], p[e], oacc[j], 0, 0, 0);
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
    if (a.key_weights && a.n_kw > a.Lself + a.Lcond) { paella_set_error("attention: attn_weights longer than the key sequence"); return PAELLA_ERR_ARG; }
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
