// VQGAN encode / decode execution plan and C ABI for gfx950 (reference src/vqgan.py:45-107).
// Same structure as model.hip: one host call enqueues the whole stack; NHWC activations in a caller-owned
// workspace; weights repacked once at load.
#include "internal.h"
#include "../../include/paella_hip.h"

#include <math.h>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

enum VqOp { VQ_CONV1, VQ_RES, VQ_CONV4S2, VQ_CONVT4, VQ_LATENT_BN };

struct VqBlock {
    VqOp op;
    int c_in, c_out;
    std::string prefix;
    float gam[6];
    DevBuf phase_w[4];  // VQ_CONVT4: [c_out, 4*c_in] per output phase
};

struct VqSpec { Repack kind; std::vector<int64_t> shape; bool is_int = false; };

struct paella_vqgan {
    paella_vqgan_config cfg;
    std::vector<int> c_levels;
    std::vector<VqBlock> enc, dec;
    std::map<std::string, VqSpec> specs;
    std::map<std::string, DevBuf> t;
    DevBuf bn_scale, bn_shift;
    bool finalized = false;
    float bn_eps = 1e-5f;
    // OPT-IN bf16 fast mode of THIS model (paella_vqgan_set_precision; outside the fp32 parity contract): bf16 shadows of the ResBlocks' MLP weights
    int precision = 0;
    std::map<std::string, DevBuf16> t16;
};

static const float* VT(const paella_vqgan* v, const std::string& k) {
    auto it = v->t.find(k);
    return it == v->t.end() ? nullptr : it->second.p;
}
static const unsigned short* VT16(const paella_vqgan* v, const std::string& k) {
    if (v->precision != 1) return nullptr;
    auto it = v->t16.find(k);
    return it == v->t16.end() ? nullptr : it->second.p;
}
static void vspec(paella_vqgan* v, const std::string& key, Repack kind, std::vector<int64_t> shape) {
    VqSpec s; s.kind = kind; s.shape = std::move(shape);
    v->specs[key] = s;
}
static void res_specs(paella_vqgan* v, const std::string& p, int64_t c) {
    vspec(v, p + ".depthwise.1.weight", RP_DW, {c, 1, 3, 3});
    vspec(v, p + ".depthwise.1.bias", RP_COPY, {c});
    vspec(v, p + ".channelwise.0.weight", RP_COPY, {4 * c, c});
    vspec(v, p + ".channelwise.0.bias", RP_COPY, {4 * c});
    vspec(v, p + ".channelwise.2.weight", RP_COPY, {c, 4 * c});
    vspec(v, p + ".channelwise.2.bias", RP_COPY, {c});
    vspec(v, p + ".gammas", RP_COPY, {6});
}

extern "C" int paella_vqgan_create(const paella_vqgan_config* cfg, paella_vqgan** out) {
    if (!cfg || !out) { paella_set_error("null argument"); return PAELLA_ERR_ARG; }
    if (cfg->levels < 1 || cfg->levels > 6 || (cfg->c_hidden >> (cfg->levels - 1)) < 8 || (cfg->c_hidden % (8 << (cfg->levels - 1))) ||
        (cfg->c_latent & 3) || cfg->c_latent > 64) {
        paella_set_error("unsupported VQGAN configuration (levels=%d c_hidden=%d c_latent=%d)", cfg->levels, cfg->c_hidden, cfg->c_latent);
        return PAELLA_ERR_ARG;
    }
    paella_vqgan* v = new paella_vqgan();
    v->cfg = *cfg;
    const int L = cfg->levels;
    for (int i = L - 1; i >= 0; --i) v->c_levels.push_back(cfg->c_hidden >> i);  // c_levels[0] smallest
    char buf[96];
    // encoder (src/vqgan.py:54-69)
    vspec(v, "in_block.1.weight", RP_COPY, {v->c_levels[0], 12, 1, 1});
    vspec(v, "in_block.1.bias", RP_COPY, {v->c_levels[0]});
    {
        VqBlock b; b.op = VQ_CONV1; b.c_in = 12; b.c_out = v->c_levels[0]; b.prefix = "in_block.1";
        v->enc.push_back(b);
    }
    int j = 0;
    for (int i = 0; i < L; ++i) {
        if (i > 0) {
            VqBlock b; b.op = VQ_CONV4S2; b.c_in = v->c_levels[i - 1]; b.c_out = v->c_levels[i];
            snprintf(buf, sizeof buf, "down_blocks.%d", j++);
            b.prefix = buf;
            vspec(v, b.prefix + ".weight", RP_CONV_K2, {b.c_out, b.c_in, 4, 4});
            vspec(v, b.prefix + ".bias", RP_COPY, {b.c_out});
            v->enc.push_back(b);
        }
        VqBlock r; r.op = VQ_RES; r.c_in = r.c_out = v->c_levels[i];
        snprintf(buf, sizeof buf, "down_blocks.%d", j++);
        r.prefix = buf;
        res_specs(v, r.prefix, r.c_in);
        v->enc.push_back(r);
    }
    {
        VqBlock b; b.op = VQ_LATENT_BN; b.c_in = v->c_levels[L - 1]; b.c_out = cfg->c_latent;
        snprintf(buf, sizeof buf, "down_blocks.%d", j);
        b.prefix = buf;
        vspec(v, b.prefix + ".0.weight", RP_COPY, {cfg->c_latent, b.c_in, 1, 1});
        vspec(v, b.prefix + ".1.weight", RP_COPY, {cfg->c_latent});
        vspec(v, b.prefix + ".1.bias", RP_COPY, {cfg->c_latent});
        vspec(v, b.prefix + ".1.running_mean", RP_COPY, {cfg->c_latent});
        vspec(v, b.prefix + ".1.running_var", RP_COPY, {cfg->c_latent});
        v->enc.push_back(b);
    }
    vspec(v, "vquantizer.codebook.weight", RP_COPY, {cfg->codebook_size, cfg->c_latent});
    // decoder (src/vqgan.py:74-89)
    {
        VqBlock b; b.op = VQ_CONV1; b.c_in = cfg->c_latent; b.c_out = v->c_levels[L - 1]; b.prefix = "up_blocks.0.0";
        vspec(v, "up_blocks.0.0.weight", RP_COPY, {b.c_out, b.c_in, 1, 1});
        vspec(v, "up_blocks.0.0.bias", RP_COPY, {b.c_out});
        v->dec.push_back(b);
    }
    j = 1;
    for (int i = 0; i < L; ++i) {
        const int cl = v->c_levels[L - 1 - i];
        const int nb = i == 0 ? cfg->bottleneck_blocks : 1;
        for (int k = 0; k < nb; ++k) {
            VqBlock r; r.op = VQ_RES; r.c_in = r.c_out = cl;
            snprintf(buf, sizeof buf, "up_blocks.%d", j++);
            r.prefix = buf;
            res_specs(v, r.prefix, cl);
            v->dec.push_back(r);
        }
        if (i < L - 1) {
            VqBlock b; b.op = VQ_CONVT4; b.c_in = cl; b.c_out = v->c_levels[L - 2 - i];
            snprintf(buf, sizeof buf, "up_blocks.%d", j++);
            b.prefix = buf;
            vspec(v, b.prefix + ".weight", RP_CONVT_K2, {b.c_in, b.c_out, 4, 4});
            vspec(v, b.prefix + ".bias", RP_COPY, {b.c_out});
            v->dec.push_back(b);
        }
    }
    vspec(v, "out_block.0.weight", RP_COPY, {12, v->c_levels[0], 1, 1});
    vspec(v, "out_block.0.bias", RP_COPY, {12});
    *out = v;
    return PAELLA_OK;
}

extern "C" void paella_vqgan_destroy(paella_vqgan* v) {
    if (!v) return;
    for (auto& kv : v->t) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& kv : v->t16) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto* seq : {&v->enc, &v->dec})
        for (auto& b : *seq)
            for (auto& pw : b.phase_w) if (pw.p) (void)hipFree(pw.p);
    if (v->bn_scale.p) (void)hipFree(v->bn_scale.p);
    if (v->bn_shift.p) (void)hipFree(v->bn_shift.p);
    delete v;
}

extern "C" int paella_vqgan_load_tensor(paella_vqgan* v, const char* key, const float* dev_src, const int64_t* shape, int ndim,
                                        void* stream) {
    if (!v || !key || !dev_src) { paella_set_error("null argument"); return PAELLA_ERR_ARG; }
    auto it = v->specs.find(key);
    if (it == v->specs.end()) { paella_set_error("unexpected state-dict key '%s'", key); return PAELLA_ERR_ARG; }
    const VqSpec& sp = it->second;
    if ((int)sp.shape.size() != ndim) { paella_set_error("%s: expected %d dims, got %d", key, (int)sp.shape.size(), ndim); return PAELLA_ERR_ARG; }
    for (int d = 0; d < ndim; ++d)
        if (shape[d] != sp.shape[d]) { paella_set_error("%s: dim %d is %lld, expected %lld", key, d, (long long)shape[d], (long long)sp.shape[d]); return PAELLA_ERR_ARG; }
    v->finalized = false;
    return repack_into(sp.kind, dev_src, sp.shape, v->t[key], (hipStream_t)stream);
}


static int vq_make_shadows(paella_vqgan* v, hipStream_t st) {
    for (auto* seq : {&v->enc, &v->dec})
        for (auto& b : *seq) {
            if (b.op != VQ_RES || (b.c_in % 64)) continue;
            for (const char* suffix : {".channelwise.0.weight", ".channelwise.2.weight"}) {
                const std::string k = b.prefix + suffix;
                auto it = v->t.find(k);
                if (it == v->t.end() || !it->second.p) continue;
                DevBuf16& d = v->t16[k];
                if (d.p && d.n != it->second.n) { (void)hipFree(d.p); d.p = nullptr; }
                if (!d.p) HIP_CHECK_RET(hipMalloc((void**)&d.p, it->second.n * sizeof(unsigned short)));
                d.n = it->second.n;
                RET_IF(launch_f32_to_bf16(it->second.p, d.p, d.n, st));
            }
        }
    HIP_CHECK_RET(hipStreamSynchronize(st));
    return PAELLA_OK;
}

extern "C" int paella_vqgan_finalize(paella_vqgan* v, void* stream) {
    if (!v) { paella_set_error("null argument"); return PAELLA_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    for (auto& kv : v->specs) {
        auto it = v->t.find(kv.first);
        if (it == v->t.end() || !it->second.loaded) { paella_set_error("tensor '%s' was never loaded", kv.first.c_str()); return PAELLA_ERR_STATE; }
    }
    HIP_CHECK_RET(hipStreamSynchronize(st));
    for (auto* seq : {&v->enc, &v->dec})
        for (auto& b : *seq) {
            if (b.op == VQ_RES) {
                HIP_CHECK_RET(hipMemcpy(b.gam, VT(v, b.prefix + ".gammas"), 6 * sizeof(float), hipMemcpyDeviceToHost));
            } else if (b.op == VQ_CONVT4) {
                // repacked weight is [ky][kx][co][ci]; phase (py,px) uses taps ky = py?{0,2}:{1,3} (ty = 0,1), same for kx
                const float* w4 = VT(v, b.prefix + ".weight");
                for (int ph = 0; ph < 4; ++ph) {
                    const int py = ph >> 1, px = ph & 1;
                    RET_IF(devbuf_alloc(b.phase_w[ph], (size_t)b.c_out * 4 * b.c_in));
                    for (int tap = 0; tap < 4; ++tap) {
                        const int ty = tap >> 1, tx = tap & 1;
                        const int ky = py == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 0 : 2);
                        const int kx = px == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 0 : 2);
                        RET_IF(launch_copy_rows(w4 + ((size_t)(ky * 4 + kx) * b.c_out) * b.c_in, b.c_in, b.phase_w[ph].p + tap * b.c_in,
                                                4 * b.c_in, b.c_out, b.c_in, st));
                    }
                }
            } else if (b.op == VQ_LATENT_BN) {
                const int cl = v->cfg.c_latent;
                std::vector<float> w(cl), bb(cl), mu(cl), var(cl), sc(cl), sh(cl);
                HIP_CHECK_RET(hipMemcpy(w.data(), VT(v, b.prefix + ".1.weight"), cl * sizeof(float), hipMemcpyDeviceToHost));
                HIP_CHECK_RET(hipMemcpy(bb.data(), VT(v, b.prefix + ".1.bias"), cl * sizeof(float), hipMemcpyDeviceToHost));
                HIP_CHECK_RET(hipMemcpy(mu.data(), VT(v, b.prefix + ".1.running_mean"), cl * sizeof(float), hipMemcpyDeviceToHost));
                HIP_CHECK_RET(hipMemcpy(var.data(), VT(v, b.prefix + ".1.running_var"), cl * sizeof(float), hipMemcpyDeviceToHost));
                for (int i = 0; i < cl; ++i) {  // eval-mode BatchNorm2d folded to y = x*sc + sh
                    const float inv = 1.0f / sqrtf(var[i] + v->bn_eps);
                    sc[i] = w[i] * inv;
                    sh[i] = bb[i] - mu[i] * sc[i];
                }
                RET_IF(devbuf_alloc(v->bn_scale, cl));
                RET_IF(devbuf_alloc(v->bn_shift, cl));
                HIP_CHECK_RET(hipMemcpy(v->bn_scale.p, sc.data(), cl * sizeof(float), hipMemcpyHostToDevice));
                HIP_CHECK_RET(hipMemcpy(v->bn_shift.p, sh.data(), cl * sizeof(float), hipMemcpyHostToDevice));
            }
        }
    HIP_CHECK_RET(hipStreamSynchronize(st));
    v->finalized = true;
    if (v->precision == 1) RET_IF(vq_make_shadows(v, st));  // (a reload refreshes the shadows)
    return PAELLA_OK;
}

// OPT-IN fast mode of THIS model: mode 1 runs the MLP of every ResBlock whose width is a multiple of 64 on bf16-operand MFMA (bf16 shadow weights, bf16
// LayerNorm output and hidden tensor; residual stream, depthwise half, the strided convolutions and the image stay fp32).  Mode 0 (default) = the exact path.
// Size workspaces (paella_vqgan_workspace_bytes) AFTER switching.
extern "C" int paella_vqgan_set_precision(paella_vqgan* v, int mode, void* stream) {
    if (!v) { paella_set_error("null argument"); return PAELLA_ERR_ARG; }
    if (mode != 0 && mode != 1) { paella_set_error("precision mode must be 0 (fp32) or 1 (bf16 operands)"); return PAELLA_ERR_ARG; }
    v->precision = mode;
    if (mode == 1 && v->finalized) return vq_make_shadows(v, (hipStream_t)stream);
    // mode 0: shadows stay allocated until paella_vqgan_destroy (see paella_unet_set_precision); no synchronisation here
    return PAELLA_OK;
}

struct VqBuffers { float *x, *t, *g, *a, *lat, *qe, *splitk; unsigned short *t16, *g16; };

// largest activation: at the image-side level the grid is (h*2^(L-1)) x (w*2^(L-1)) with c_levels[0] channels
static void vq_carve(const paella_vqgan* v, Arena& a, int B, int h, int w, VqBuffers& f) {
    f.splitk = a.take(kSplitKBudget / sizeof(float));  // FIRST: its ticket header sits at a fixed offset (paella_workspace_init)
    const int L = v->cfg.levels;
    size_t xmax = 0, amax = 0;
    for (int i = 0; i < L; ++i) {
        const size_t rows = (size_t)B * (h << (L - 1 - i)) * (w << (L - 1 - i));  // rows at encoder level i
        const size_t n = rows * v->c_levels[i];
        if (n > xmax) xmax = n;
        // gather operands: convT phase A at (level i+1 rows) x 4*c_levels[i+1]; im2col at (level i+1 rows) x 16*c_levels[i]
        if (i + 1 < L) {
            const size_t r1 = (size_t)B * (h << (L - 2 - i)) * (w << (L - 2 - i));
            const size_t a1 = r1 * 4 * v->c_levels[i + 1], a2 = r1 * 16 * v->c_levels[i];
            if (a1 > amax) amax = a1;
            if (a2 > amax) amax = a2;
        }
    }
    const size_t img_rows = (size_t)B * (h << (L - 1)) * (w << (L - 1));
    if (img_rows * 12 > amax) amax = img_rows * 12;
    f.x = a.take(xmax);
    f.t = a.take(xmax);
    f.g = a.take(4 * xmax);
    f.a = a.take(amax ? amax : 4);
    f.lat = a.take((size_t)B * h * w * v->cfg.c_latent);
    f.qe = a.take((size_t)B * h * w * v->cfg.c_latent);
    f.t16 = f.g16 = nullptr;
    if (v->precision == 1) {  // bf16 fast mode: LayerNorm output and hidden tensor of the ResBlock MLPs (after everything else: the fp32 layout does not move)
        f.t16 = reinterpret_cast<unsigned short*>(a.take(xmax / 2 + 64));
        f.g16 = reinterpret_cast<unsigned short*>(a.take(2 * xmax + 64));
    }
}

extern "C" size_t paella_vqgan_workspace_bytes(const paella_vqgan* v, int B, int h, int w) {
    if (!v) return 0;
    VqBuffers f;
    Arena a(nullptr, 0);
    vq_carve(v, a, B, h, w, f);
    return a.off + 256;
}

// ResBlock (reference src/vqgan.py:34-42); x updated in place
static int vq_resblock(const paella_vqgan* v, const VqBlock& b, VqBuffers& f, int B, int h, int w, hipStream_t st) {
    const int c = b.c_in;
    const int64_t rows = (int64_t)B * h * w;
    RET_IF(launch_layernorm(f.x, f.t, rows, c, 1e-6f, 1.0f + b.gam[0], b.gam[1], 0, 0, 0, st));
    RET_IF(launch_dwconv_res(f.x, f.t, VT(v, b.prefix + ".depthwise.1.weight"), VT(v, b.prefix + ".depthwise.1.bias"), f.x, B, h, w, c,
                             b.gam[2], st));
    const unsigned short* const w1_16 = VT16(v, b.prefix + ".channelwise.0.weight");
    const unsigned short* const w2_16 = VT16(v, b.prefix + ".channelwise.2.weight");
    if (w1_16 && w2_16 && f.t16 && (c % 64) == 0) {  // opt-in bf16 fast mode: LayerNorm -> bf16 | GEMM1 -> GELU -> bf16 hidden | GEMM2 -> fp32 residual stream
        RET_IF(launch_layernorm16(f.x, nullptr, f.t16, rows, c, 1e-6f, 1.0f + b.gam[3], b.gam[4], 0, 0, 0, st));
        GemmArgs g1 = gemm_args(nullptr, c, VT(v, b.prefix + ".channelwise.0.weight"), c, nullptr, 4 * c, (int)rows, 4 * c, c);
        g1.A16 = f.t16; g1.W16 = w1_16;
        g1.ep.bias = VT(v, b.prefix + ".channelwise.0.bias");
        g1.ep.act = ACT_GELU;
        g1.ep.c16 = f.g16;
        RET_IF(launch_gemm(g1, f.splitk, kSplitKBudget, st));
        GemmArgs g2 = gemm_args(nullptr, 4 * c, VT(v, b.prefix + ".channelwise.2.weight"), 4 * c, f.x, c, (int)rows, c, 4 * c);
        g2.A16 = f.g16; g2.W16 = w2_16;
        g2.ep.bias = VT(v, b.prefix + ".channelwise.2.bias");
        g2.ep.alpha = b.gam[5];
        g2.ep.residual = f.x; g2.ep.ldr = c;
        RET_IF(launch_gemm(g2, f.splitk, kSplitKBudget, st));
        return PAELLA_OK;
    }
    RET_IF(launch_layernorm(f.x, f.t, rows, c, 1e-6f, 1.0f + b.gam[3], b.gam[4], 0, 0, 0, st));
    GemmArgs g1 = gemm_args(f.t, c, VT(v, b.prefix + ".channelwise.0.weight"), c, f.g, 4 * c, (int)rows, 4 * c, c);
    g1.ep.bias = VT(v, b.prefix + ".channelwise.0.bias");
    g1.ep.act = ACT_GELU;
    RET_IF(launch_gemm(g1, f.splitk, kSplitKBudget, st));
    GemmArgs g2 = gemm_args(f.g, 4 * c, VT(v, b.prefix + ".channelwise.2.weight"), 4 * c, f.x, c, (int)rows, c, 4 * c);
    g2.ep.bias = VT(v, b.prefix + ".channelwise.2.bias");
    g2.ep.alpha = b.gam[5];
    g2.ep.residual = f.x; g2.ep.ldr = c;
    RET_IF(launch_gemm(g2, f.splitk, kSplitKBudget, st));
    return PAELLA_OK;
}

static int vq_run_decoder(paella_vqgan* v, VqBuffers& f, int B, int h, int w, float* img_out, hipStream_t st) {
    // f.lat holds the [B*h*w, c_latent] decoder input
    int ch = h, cw = w;
    const float* in = f.lat;
    for (VqBlock& b : v->dec) {
        const int64_t rows = (int64_t)B * ch * cw;
        switch (b.op) {
            case VQ_CONV1: {
                GemmArgs g = gemm_args(in, b.c_in, VT(v, b.prefix + ".weight"), b.c_in, f.x, b.c_out, (int)rows, b.c_out, b.c_in);
                g.ep.bias = VT(v, b.prefix + ".bias");
                RET_IF(launch_gemm(g, f.splitk, kSplitKBudget, st));
                break;
            }
            case VQ_RES: RET_IF(vq_resblock(v, b, f, B, ch, cw, st)); break;
            case VQ_CONVT4: {  // 4 output phases, each a gather + GEMM with a strided (depth-to-space) store into f.t
                for (int ph = 0; ph < 4; ++ph) {
                    const int py = ph >> 1, px = ph & 1;
                    // implicit GEMM: the 2x2 taps of this phase are gathered in the GEMM's operand load (no [rows, 4*c_in] buffer);
                    // channel counts that are not a multiple of the K step (tiny test models) take the materialised operand
                    const bool implicit = (b.c_in & 31) == 0;
                    if (!implicit) RET_IF(launch_convT4_gather(f.x, f.a, B, ch, cw, b.c_in, py, px, st));
                    GemmArgs g = gemm_args(implicit ? f.x : f.a, 4 * b.c_in, b.phase_w[ph].p, 4 * b.c_in, f.t, b.c_out, (int)rows, b.c_out, 4 * b.c_in);
                    if (implicit) {
                        g.cv.enabled = 1; g.cv.Hi = ch; g.cv.Wi = cw; g.cv.C = b.c_in; g.cv.Ho = ch; g.cv.Wo = cw; g.cv.stride = 1; g.cv.ntaps = 4;
                        // tap (ty, tx) of output phase (py, px) reads input (y + py - ty, x + px - tx): src/vqgan.py:83-85 unrolled per phase
                        g.cv.tw_log2 = 1; g.cv.oy0 = py; g.cv.ox0 = px; g.cv.tsign = -1;
                    }
                    g.ep.bias = VT(v, b.prefix + ".bias");
                    g.ep.store_mode = STORE_D2S; g.ep.sH = ch; g.ep.sW = cw; g.ep.sC = b.c_out; g.ep.n_seg_x = 1; g.ep.py = py; g.ep.px = px;
                    RET_IF(launch_gemm(g, f.splitk, kSplitKBudget, st));
                }
                float* tmp = f.x; f.x = f.t; f.t = tmp;
                ch <<= 1; cw <<= 1;
                break;
            }
            default: break;
        }
    }
    // out_block: Conv1x1 -> 12, PixelShuffle(2) -> NCHW image
    const int64_t rows = (int64_t)B * ch * cw;
    GemmArgs g = gemm_args(f.x, v->c_levels[0], VT(v, "out_block.0.weight"), v->c_levels[0], img_out, 4, (int)rows, 12, v->c_levels[0]);
    g.ep.bias = VT(v, "out_block.0.bias");
    g.ep.store_mode = STORE_PIXSHUF_NCHW; g.ep.sH = ch; g.ep.sW = cw; g.ep.sC = 3;
    RET_IF(launch_gemm(g, f.splitk, kSplitKBudget, st));
    return PAELLA_OK;
}

static int vq_prepare(paella_vqgan* v, VqBuffers& f, int B, int h, int w, void* ws, size_t ws_bytes) {
    if (!v || !v->finalized) { paella_set_error("VQGAN not finalized"); return PAELLA_ERR_STATE; }
    if (B <= 0 || h <= 0 || w <= 0) { paella_set_error("bad latent grid"); return PAELLA_ERR_ARG; }
    Arena a(ws, ws_bytes);
    vq_carve(v, a, B, h, w, f);
    if (!a.ok || !ws) { paella_set_error("workspace too small (%zu needed, %zu given)", a.off, ws_bytes); return PAELLA_ERR_WORKSPACE; }
    return PAELLA_OK;
}

extern "C" int paella_vqgan_decode_indices(paella_vqgan* v, const int64_t* idx, int B, int h, int w, float* img_out, void* ws,
                                           size_t ws_bytes, void* stream) {
    VqBuffers f;
    RET_IF(vq_prepare(v, f, B, h, w, ws, ws_bytes));
    hipStream_t st = (hipStream_t)stream;
    RET_IF(launch_codebook_gather(idx, VT(v, "vquantizer.codebook.weight"), f.lat, (int64_t)B * h * w, v->cfg.c_latent, v->cfg.codebook_size, 1.0f, st));
    return vq_run_decoder(v, f, B, h, w, img_out, st);
}

extern "C" int paella_vqgan_decode(paella_vqgan* v, const float* latents, int B, int h, int w, float* img_out, void* ws, size_t ws_bytes,
                                   void* stream) {
    VqBuffers f;
    RET_IF(vq_prepare(v, f, B, h, w, ws, ws_bytes));
    hipStream_t st = (hipStream_t)stream;
    RET_IF(launch_nchw_to_nhwc(latents, f.lat, B, h * w, v->cfg.c_latent, v->cfg.scale_factor, 0, st));
    return vq_run_decoder(v, f, B, h, w, img_out, st);
}

__global__ __launch_bounds__(256) void vq_loss_kernel(const float* __restrict__ qe, const float* __restrict__ x, int64_t n, float* __restrict__ out, int combined) {
    // single workgroup, fixed-order reduction: mse = mean((qe - x)^2); combined: out = vq_loss + 0.25 * commit_loss = 1.25 * mse
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double d = (double)qe[i] - (double)x[i];
        s += d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = (float)(red[0] / (double)n);
        out[0] = combined ? mse + mse * 0.25f : mse;
    }
}

extern "C" int paella_vqgan_encode(paella_vqgan* v, const float* img, int B, int Hp, int Wp, float* qe_out, float* x_out, int64_t* idx_out,
                                   float* loss_out, void* ws, size_t ws_bytes, void* stream) {
    if (!v) { paella_set_error("null argument"); return PAELLA_ERR_ARG; }
    const int L = v->cfg.levels;
    const int f_ = 1 << L;
    if (!img || Hp % f_ || Wp % f_) { paella_set_error("image size %dx%d must be a multiple of %d", Hp, Wp, f_); return PAELLA_ERR_ARG; }
    const int h = Hp / f_, w = Wp / f_;
    VqBuffers f;
    RET_IF(vq_prepare(v, f, B, h, w, ws, ws_bytes));
    hipStream_t st = (hipStream_t)stream;
    int ch = Hp / 2, cw = Wp / 2;
    RET_IF(launch_img_unshuffle(img, f.a, B, 3, Hp, Wp, st));
    for (VqBlock& b : v->enc) {
        const int64_t rows = (int64_t)B * ch * cw;
        switch (b.op) {
            case VQ_CONV1: {
                GemmArgs g = gemm_args(f.a, 12, VT(v, b.prefix + ".weight"), 12, f.x, b.c_out, (int)rows, b.c_out, 12);
                g.ep.bias = VT(v, b.prefix + ".bias");
                RET_IF(launch_gemm(g, f.splitk, kSplitKBudget, st));
                break;
            }
            case VQ_RES: RET_IF(vq_resblock(v, b, f, B, ch, cw, st)); break;
            case VQ_CONV4S2: {
                const bool implicit = (b.c_in & 31) == 0;  // implicit GEMM (no im2col buffer) whenever the K step divides the channel count
                if (!implicit) RET_IF(launch_conv4s2_im2col(f.x, f.a, B, ch, cw, b.c_in, st));
                GemmArgs g = gemm_args(implicit ? f.x : f.a, 16 * b.c_in, VT(v, b.prefix + ".weight"), 16 * b.c_in, f.t, b.c_out, (int)(rows / 4), b.c_out, 16 * b.c_in);
                if (implicit) {
                    g.cv.enabled = 1; g.cv.Hi = ch; g.cv.Wi = cw; g.cv.C = b.c_in; g.cv.Ho = ch / 2; g.cv.Wo = cw / 2; g.cv.stride = 2; g.cv.ntaps = 16;
                    g.cv.tw_log2 = 2; g.cv.oy0 = -1; g.cv.ox0 = -1; g.cv.tsign = 1;  // Conv2d(k4, s2, p1): tap (ky, kx) reads (2*yo - 1 + ky, 2*xo - 1 + kx), src/vqgan.py:61
                }
                ch >>= 1; cw >>= 1;
                g.ep.bias = VT(v, b.prefix + ".bias");
                RET_IF(launch_gemm(g, f.splitk, kSplitKBudget, st));
                float* tmp = f.x; f.x = f.t; f.t = tmp;
                break;
            }
            case VQ_LATENT_BN: {
                const int cl = v->cfg.c_latent;
                GemmArgs g = gemm_args(f.x, b.c_in, VT(v, b.prefix + ".0.weight"), b.c_in, f.t, cl, (int)rows, cl, b.c_in);
                RET_IF(launch_gemm(g, f.splitk, kSplitKBudget, st));
                RET_IF(launch_affine_cols(f.t, v->bn_scale.p, v->bn_shift.p, f.lat, rows, cl, st));
                break;
            }
            default: break;
        }
    }
    const int cl = v->cfg.c_latent;
    const int64_t rows = (int64_t)B * h * w;
    int64_t* idx = idx_out ? idx_out : (int64_t*)f.g;  // scratch when the caller does not want indices
    RET_IF(launch_vq_nearest(f.lat, VT(v, "vquantizer.codebook.weight"), idx, f.qe, rows, cl, v->cfg.codebook_size, st));
    // reference returns qe / scale_factor and x / scale_factor (true divisions)
    if (qe_out) RET_IF(launch_nhwc_to_nchw(f.qe, qe_out, B, h * w, cl, v->cfg.scale_factor, 1, st));
    if (x_out) RET_IF(launch_nhwc_to_nchw(f.lat, x_out, B, h * w, cl, v->cfg.scale_factor, 1, st));
    if (loss_out) {
        hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(256), 0, st, f.qe, f.lat, rows * cl, loss_out, 1);
        LAUNCH_CHECK_RET();
    }
    return PAELLA_OK;
}

extern "C" int paella_vqgan_quantize_rows(paella_vqgan* v, const float* x, int64_t rows, int64_t* idx_out, float* qe_out, float* mse_out,
                                          void* stream) {
    if (!v || !v->finalized) { paella_set_error("VQGAN not finalized"); return PAELLA_ERR_STATE; }
    if (!x || !idx_out || (mse_out && !qe_out)) { paella_set_error("null argument"); return PAELLA_ERR_ARG; }
    RET_IF(launch_vq_nearest(x, VT(v, "vquantizer.codebook.weight"), idx_out, qe_out, rows, v->cfg.c_latent, v->cfg.codebook_size, (hipStream_t)stream));
    if (mse_out && rows > 0) {
        hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, qe_out, x, rows * v->cfg.c_latent, mse_out, 0);
        LAUNCH_CHECK_RET();
    }
    return PAELLA_OK;
}

// VectorQuantize.idx2vq (used at src/vqgan.py:104): codebook rows for a flat index list, out fp32 [rows, c_latent]
extern "C" int paella_vqgan_lookup_rows(paella_vqgan* v, const int64_t* idx, int64_t rows, float* out, void* stream) {
    if (!v || !v->finalized) { paella_set_error("VQGAN not finalized"); return PAELLA_ERR_STATE; }
    if (!idx || !out) { paella_set_error("null argument"); return PAELLA_ERR_ARG; }
    return launch_codebook_gather(idx, VT(v, "vquantizer.codebook.weight"), out, rows, v->cfg.c_latent, v->cfg.codebook_size, 1.0f, (hipStream_t)stream);
}
