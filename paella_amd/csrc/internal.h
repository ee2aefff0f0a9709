// Host-side helpers shared by model.hip (UNet) and vqmodel.hip (VQGAN).
#pragma once
#include "common.h"
#include <vector>

#define RET_IF(expr)                      \
    do {                                  \
        int _rc = (expr);                 \
        if (_rc != PAELLA_OK) return _rc; \
    } while (0)

// library-owned device tensor (repacked weight)
struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    bool loaded = false;
};
int devbuf_alloc(DevBuf& b, size_t n);
struct DevBuf16 {  // bf16 shadow copy of a weight (opt-in fast mode)
    unsigned short* p = nullptr;
    size_t n = 0;
};

// bump allocator over a caller-owned workspace; base == nullptr -> size query only
struct Arena {
    char* base;
    size_t cap, off;
    bool ok;
    Arena(void* b, size_t c) : base((char*)b), cap(c), off(0), ok(true) {}
    float* take(size_t nfloats) {
        const size_t bytes = (nfloats * sizeof(float) + 255) & ~(size_t)255;
        const size_t o = off;
        off += bytes;
        if (base && off > cap) ok = false;
        return base ? (float*)(base + o) : nullptr;
    }
};

static const size_t kSplitKBudget = (size_t)96 << 20;  // bytes of the split-K region (ticket header + slabs) at the START of every workspace

enum Repack {
    RP_COPY,      // as is
    RP_DW,        // depthwise [C, J, 3, 3] -> [J, 3, 3, C]
    RP_CONV_K2,   // conv [co, ci, kh, kw] -> [co, kh, kw, ci]
    RP_CONVT_K2,  // transposed conv [ci, co, kh, kw] -> [kh, kw, co, ci]
    RP_TILE4,     // bias [c] -> [4][c]
    RP_CLF_W, RP_CLF_B, RP_TS_W, RP_TS_B  // handled by the UNet loader
};
int repack_into(Repack kind, const float* src, const std::vector<int64_t>& shape, DevBuf& dst, hipStream_t st);

static inline GemmArgs gemm_args(const float* A, int lda, const float* Wt, int ldw, float* C, int ldc, int M, int N, int K) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.a_scale = nullptr; g.a_shift = nullptr; g.a_rows_per_sample = 1; g.a_rps_div.mul = 0; g.a_rps_div.shr = 0; g.a_rps_div.pass = 0xffffffffu; g.grn_gx = nullptr; g.grn_gamma = nullptr; g.grn_part = nullptr; g.grn_np = 0; g.force_ring_cfg = 0; g.ln_stats = nullptr; g.ln_nblk = 0; g.ln_eps = 1e-6f; g.ln_wsum = nullptr; g.ln_row = nullptr; g.ln_fold_ratio = 4.0f; g.ln_guard_count = nullptr;
    g.ep = make_epilogue();
    g.ft.temperature = 1.f; g.ft.mode = 0; g.ft.seed = 0; g.ft.seed_ptr = nullptr; g.ft.offset = 0; g.ft.row_offset = 0; g.ft.row_offset_ptr = nullptr;
    g.ft.part_score = nullptr; g.ft.part_idx = nullptr;
    g.cv.enabled = 0;
    g.A16 = nullptr; g.W16 = nullptr;
    return g;
}
