"""Sweep (tile config, split-K) of the fp32 MFMA GEMM over the shapes of the sampling path and print TFLOP/s per variant.
Usage (GPU box): python tools/gemm_tune.py [--out gpurun_out/gemm_tune.json]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paella_amd import _lib

# (M, N, K, apro) for the 570M-class model at 32x32 tokens, cond+uncond batched (2 rows), plus batch-8 variants
SHAPES = {
    "L0 mlp1 512x2560x640": (512, 2560, 640), "L0 mlp2 512x640x2560": (512, 640, 2560),
    "L1 mlp1 128x5120x1280": (128, 5120, 1280), "L1 mlp2 128x1280x5120": (128, 1280, 5120),
    "L1 qkv 128x3840x1280": (128, 3840, 1280), "L1 out 128x1280x1280": (128, 1280, 1280),
    "L2 mlp1 32x5120x1280": (32, 5120, 1280), "L2 mlp2 32x1280x5120": (32, 1280, 5120),
    "L2 qkv 32x3840x1280": (32, 3840, 1280), "L2 out 32x1280x1280": (32, 1280, 1280),
    "embed 512x640x1024": (512, 640, 1024), "down1 128x1280x2560": (128, 1280, 2560), "down2 32x1280x5120": (32, 1280, 5120),
    "up2 32x5120x1280": (32, 5120, 1280), "up1 128x2560x1280": (128, 2560, 1280), "clf 512x1024x640": (512, 1024, 640),
    "out 2048x8192x256": (2048, 8192, 256),
    # shared classifier-free-guidance prefix (distinct rows only) and the folded head
    "pfx L0 mlp1 256x2560x640": (256, 2560, 640), "pfx L0 mlp2 256x640x2560": (256, 640, 2560), "pfx embed 256x640x1024": (256, 640, 1024),
    "pfx down1 64x1280x2560": (64, 1280, 2560), "pfx L1 mlp1 64x5120x1280": (64, 5120, 1280), "pfx L1 mlp2 64x1280x5120": (64, 1280, 5120),
    "head 1024x8192x256": (1024, 8192, 256),
    # BASELINE configs[2] (batch 64, 64x64 tokens, cond+uncond rows): the MFMA-bound regime
    "c3 L0 mlp1 131072x2560x640": (131072, 2560, 640), "c3 L0 mlp2 131072x640x2560": (131072, 640, 2560),
    "c3 L1 mlp1 32768x5120x1280": (32768, 5120, 1280), "c3 L1 mlp2 32768x1280x5120": (32768, 1280, 5120),
    "c3 L1 qkv 32768x3840x1280": (32768, 3840, 1280), "c3 L2 mlp1 8192x5120x1280": (8192, 5120, 1280),
    "c3 head 262144x8192x256": (262144, 8192, 256),
    # batch 32 at 32x32 tokens
    "b32 L1 mlp1 4096x5120x1280": (4096, 5120, 1280), "b32 L1 mlp2 4096x1280x5120": (4096, 1280, 5120), "b32 L1 qkv 4096x3840x1280": (4096, 3840, 1280),
    "b32 L1 out 4096x1280x1280": (4096, 1280, 1280), "b32 L0 mlp1 16384x2560x640": (16384, 2560, 640), "b32 L2 mlp1 1024x5120x1280": (1024, 5120, 1280),
    "b32 L2 mlp2 1024x1280x5120": (1024, 1280, 5120),
    "b32 L0 mlp2 16384x640x2560": (16384, 640, 2560), "b32 L2 qkv 1024x3840x1280": (1024, 3840, 1280), "b32 L2 out 1024x1280x1280": (1024, 1280, 1280),
    "c3 L2 mlp2 8192x1280x5120": (8192, 1280, 5120), "c3 L2 out 8192x1280x1280": (8192, 1280, 1280), "c3 L2 qkv 8192x3840x1280": (8192, 3840, 1280),
    "b32 pfx L0 mlp1 8192x2560x640": (8192, 2560, 640), "b32 pfx L0 mlp2 8192x640x2560": (8192, 640, 2560), "b32 clf 16384x1024x640": (16384, 1024, 640),
    "b8 L0 mlp1 4096x2560x640": (4096, 2560, 640), "b8 L0 mlp2 4096x640x2560": (4096, 640, 2560),
    "b8 L1 mlp1 1024x5120x1280": (1024, 5120, 1280), "b8 L1 mlp2 1024x1280x5120": (1024, 1280, 5120),
    "b8 L2 mlp1 256x5120x1280": (256, 5120, 1280), "b8 L2 mlp2 256x1280x5120": (256, 1280, 5120),
    "vq mlp1 1024x1536x384": (1024, 1536, 384), "vq mlp2 1024x384x1536": (1024, 384, 1536),
    "vq mlp1 16384x384x96": (16384, 384, 96), "vq mlp2 16384x96x384": (16384, 96, 384),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--big-only", action="store_true", help="only the large-tile candidates without split-K")
    ap.add_argument("--only", default=None, help="substring filter on the shape name (comma-separated alternatives)")
    ap.add_argument("--apro", type=int, default=0, choices=[0, 1, 2], help="A-operand prologue: 0 none, 1 GRN scale/shift (rows per sample = 64), 2 LayerNorm from row statistics")
    ap.add_argument("--raster", type=int, default=None, help="tile rows per rasterisation group (test hook; 0 = plain m-fastest order)")
    ap.add_argument("--copies", type=int, default=0, help="rotate over exactly this many weight copies (2-3 = the weights stay in the 256 MiB Infinity Cache: "
                    "the upper bound of what a weight prefetch could buy); default: enough copies to exceed it (cold weights, as in the model)")
    ap.add_argument("--act", type=int, default=0, help="1 = bias + GELU(erf) epilogue (the MLP's first GEMM)")
    ap.add_argument("--cfgs", default=None, help="comma-separated tile configs to sweep (default: all that fit)")
    a = ap.parse_args()
    lib = _lib.load()
    if a.raster is not None:
        lib.paella_test_gemm_raster(a.raster)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = _lib.new_workspace(256 << 20, "cuda")
    results = {}
    for name, (M, N, K) in SHAPES.items():
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        # cold weights without a flush kernel: rotate over enough distinct copies of W to exceed the 256 MiB MALL,
        # exactly like consecutive layers of the model; activations stay warm.  One event pair brackets a whole
        # rotation of back-to-back launches, so the figure includes the real launch boundaries.
        ncopy = max(2, min(64, int(600e6 // (N * K * 4)) + 1)) if M < 4096 else 3
        if a.copies:
            ncopy = a.copies
        A = torch.randn(M, K, device="cuda")
        Ws = [torch.randn(N, K, device="cuda") for _ in range(ncopy)]
        bias = torch.randn(N, device="cuda") if a.act else None
        C = torch.empty(M, N, device="cuda")
        rps = 64 if M % 64 == 0 else 16
        scale = torch.ones(M // rps, K, device="cuda")
        shift = torch.zeros(K, device="cuda")
        stats = torch.stack([torch.zeros(M, K // 16, device="cuda"), torch.full((M, K // 16), 16.0, device="cuda")], dim=-1).contiguous()
        row = {}
        # (tile config, splitk): splitk > 0 = tiles * splitk workgroups (classic split-K), splitk < 0 = exactly -splitk
        # workgroups walking balanced contiguous (tile, K-step) ranges
        tile_of = {0: (128, 128), 1: (128, 64), 2: (64, 64), 3: (64, 32), 4: (32, 64), 5: (32, 32), 6: (16, 64), 7: (16, 128), 8: (32, 128),
                   9: (128, 128), 10: (128, 128), 11: (128, 32), 12: (128, 64), 13: (64, 64), 14: (128, 64), 15: (256, 32), 16: (256, 64),
                   17: (128, 64), 18: (64, 64), 19: (32, 32), 20: (128, 64), 21: (64, 32), 22: (32, 128), 23: (32, 64),
                   24: (32, 32), 25: (32, 32), 26: (64, 64), 27: (128, 32), 28: (32, 64), 29: (128, 64),
                   30: (32, 32), 31: (32, 32), 32: (32, 64), 33: (64, 32), 34: (64, 64), 35: (32, 64), 36: (256, 128)}  # 30..35: LDS-DMA ring tiles; 36: 256x128 on 8 waves
        bk_of = {c: (64 if 24 <= c <= 29 else 32) for c in tile_of}
        def fits(c):  # skip tiles that waste more than half their rows on this M
            bm = tile_of[c][0]
            return bm <= 2 * max(M, 16) or c in (2, 5)
        cands = [c for c in tile_of if fits(c) and (a.cfgs is None or str(c) in a.cfgs.split(","))]
        variants = [(-1, 1)] + [(c, s) for c in cands for s in (1, 2, 3, 4, 6, 8, 12, 16)] + \
                   [(c, -G) for c in cands for G in (256, 384, 512, 640, 768, 1024, 1280, 1536, 2048)]
        if a.big_only:
            variants = [(-1, 1)] + [(c, 1) for c in (0, 1, 2, 9, 10, 12, 14, 16, 18)]
        if M >= 4096:  # MFMA-bound: no split-K, few candidates, few iterations
            big = (0, 1, 2, 9, 10, 12, 14, 16, 18, 36) + tuple(c for c in (32, 33, 34) if a.cfgs and str(c) in a.cfgs.split(","))
            if a.cfgs:
                big = tuple(c for c in big if str(c) in a.cfgs.split(","))
            variants = [(-1, 1)] + [(c, 1) for c in big] + \
                       [(c, -G) for c in big if c in (2, 10, 14, 18) or c >= 30 for G in ((256, 512) if c == 36 else (256, 512, 768, 1024))]
        for cfg, sk in variants:
            if sk > 1 and K // sk < 96:
                continue
            if sk < 0:
                bm, bn = tile_of[cfg]
                T = -(-M // bm) * -(-N // bn)
                U = T * -(-K // bk_of[cfg])
                if -sk > U or U / -sk < 2.5 or -sk < T // 4:   # too few units per workgroup / more than 4 tiles per workgroup
                    continue
            def run(W):
                if a.apro:
                    return lib.paella_test_gemm_prologue(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, a.apro, scale.data_ptr(), shift.data_ptr(), rps,
                                                         stats.data_ptr(), cfg, sk, ws.data_ptr(), ws.numel(), st())
                return lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr() if a.act else None, None, C.data_ptr(), M, N, K, a.act, cfg, sk, ws.data_ptr(), ws.numel(), st())
            if run(Ws[0]) != 0:
                continue
            ts = []
            for _ in range(max(3, a.iters // 6) if M < 4096 else 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for W in Ws:
                    run(W)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / ncopy)
            ts.sort()
            us = ts[len(ts) // 2]
            row["%d/%d" % (cfg, sk)] = round(us, 2)
        flops = 2.0 * M * N * K
        best = min((v, k) for k, v in row.items() if not k.startswith("-1"))
        top = sorted((v, k) for k, v in row.items() if not k.startswith("-1"))[:6]
        heur = row.get("-1/1")
        results[name] = {"MNK": [M, N, K], "us": row, "best": best[1], "best_us": best[0], "best_tflops": round(flops / best[0] / 1e6, 1),
                         "heuristic_us": heur, "heuristic_tflops": round(flops / heur / 1e6, 1) if heur else None,
                         "hbm_floor_us": round((M * K + N * K + M * N) * 4 / 6.3e6, 2), "mfma_floor_us": round(flops / 157.3e6, 2)}
        r = results[name]
        print("%-34s all %s" % (name, " ".join("%s=%.0f" % kv for kv in row.items())) if (M >= 4096 or a.big_only) else "", end="\n" if (M >= 4096 or a.big_only) else "")
        print("%-28s best %-8s %8.1f us %6.1f TF | heuristic %8.1f us %6.1f TF | floors hbm %.1f mfma %.1f us | top: %s" %
              (name, r["best"], r["best_us"], r["best_tflops"], heur, r["heuristic_tflops"], r["hbm_floor_us"], r["mfma_floor_us"],
               " ".join("%s=%.1f" % (k, v) for v, k in top)), flush=True)
        ring = sorted((v, k) for k, v in row.items() if int(k.split("/")[0]) >= 30)[:4]
        legacy = sorted((v, k) for k, v in row.items() if 0 <= int(k.split("/")[0]) < 30)[:3]
        if ring and legacy:
            print("%-28s   ring tiles: %s | register-staged / 1-deep tiles: %s" % ("", " ".join("%s=%.1f" % (k, v) for v, k in ring), " ".join("%s=%.1f" % (k, v) for v, k in legacy)), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(results, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
