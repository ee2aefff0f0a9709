"""Sweep (tile config, workgroup count) of the bf16-operand instantiations of gemm_nt_kernel (the opt-in fast mode) over the shapes of the
sampling path: what choose_config_bf16 in paella_amd/csrc/gemm.hip is fitted to.
Usage (GPU box): python tools/gemm_tune_bf16.py [--only c3,b32] [--ln] [--out gpurun_out/gemm_tune_bf16.json]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paella_amd import _lib

SHAPES = {
    # batch 1 at 32x32 tokens (cond + uncond rows)
    "b1 L0 mlp1 512x2560x640": (512, 2560, 640), "b1 L0 mlp2 512x640x2560": (512, 640, 2560),
    "b1 L1 mlp1 128x5120x1280": (128, 5120, 1280), "b1 L1 mlp2 128x1280x5120": (128, 1280, 5120),
    "b1 L1 qkv 128x3840x1280": (128, 3840, 1280), "b1 L1 out 128x1280x1280": (128, 1280, 1280),
    "b1 L2 mlp1 32x5120x1280": (32, 5120, 1280), "b1 L2 mlp2 32x1280x5120": (32, 1280, 5120), "b1 L2 out 32x1280x1280": (32, 1280, 1280),
    "b1 head 1024x8192x256": (1024, 8192, 256),
    # batch 32 at 32x32 tokens
    "b32 L0 mlp1 16384x2560x640": (16384, 2560, 640), "b32 L0 mlp2 16384x640x2560": (16384, 640, 2560),
    "b32 L1 mlp1 4096x5120x1280": (4096, 5120, 1280), "b32 L1 mlp2 4096x1280x5120": (4096, 1280, 5120),
    "b32 L1 qkv 4096x3840x1280": (4096, 3840, 1280), "b32 L1 out 4096x1280x1280": (4096, 1280, 1280),
    "b32 L2 mlp1 1024x5120x1280": (1024, 5120, 1280), "b32 L2 mlp2 1024x1280x5120": (1024, 1280, 5120), "b32 L2 out 1024x1280x1280": (1024, 1280, 1280),
    "b32 head 32768x8192x256": (32768, 8192, 256),
    # BASELINE configs[2] (batch 64, 64x64 tokens, cond + uncond rows)
    "c3 L0 mlp1 131072x2560x640": (131072, 2560, 640), "c3 L0 mlp2 131072x640x2560": (131072, 640, 2560),
    "c3 L1 mlp1 32768x5120x1280": (32768, 5120, 1280), "c3 L1 mlp2 32768x1280x5120": (32768, 1280, 5120),
    "c3 L1 qkv 32768x3840x1280": (32768, 3840, 1280), "c3 L1 out 32768x1280x1280": (32768, 1280, 1280),
    "c3 L2 mlp1 8192x5120x1280": (8192, 5120, 1280), "c3 L2 mlp2 8192x1280x5120": (8192, 1280, 5120),
}
TILE = {10: (128, 128), 18: (64, 64), 19: (32, 32), 30: (32, 32), 31: (32, 32), 32: (32, 64), 33: (64, 32), 34: (64, 64), 35: (32, 64), 36: (256, 128), 37: (256, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="substring filter on the shape name (comma-separated alternatives)")
    ap.add_argument("--ln", action="store_true", help="LayerNorm folded into the epilogue (row statistics operand)")
    ap.add_argument("--act", type=int, default=0, help="1 = bias + GELU epilogue and a bf16-only output (the MLP's first GEMM)")
    ap.add_argument("--no-store", action="store_true", help="no output at all (C and C16 null): main loop + ramp only -- what the epilogue's stores cost is the difference")
    a = ap.parse_args()
    lib = _lib.load()
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = _lib.new_workspace(256 << 20, "cuda")
    results = {}
    for name, (M, N, K) in SHAPES.items():
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        ncopy = max(2, min(64, int(600e6 // (N * K * 2)) + 1)) if M < 4096 else 3   # cold weights: rotate over more copies than the Infinity Cache holds
        A = torch.randn(M, K, device="cuda").bfloat16()
        Ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(ncopy)]
        bias = torch.randn(N, device="cuda") if a.act else None
        C = torch.empty(M, N, device="cuda") if not (a.act or a.no_store) else None
        C16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if a.act else None
        stats = torch.stack([torch.zeros(M, K // 16, device="cuda"), torch.full((M, K // 16), 16.0, device="cuda")], dim=-1).contiguous() if a.ln else None
        big = M >= 4096
        variants = [(-1, 1)]
        for c, (bm, bn) in TILE.items():
            T = -(-M // bm) * -(-N // bn)
            U = T * (K // 64)
            if big:
                if c in (19, 30, 31, 32, 33, 35):
                    continue
                variants.append((c, 1))
                for Gw in (256, 512):
                    if Gw < T and U / Gw >= 2.5:
                        variants.append((c, -Gw))
            else:
                if bm > 2 * max(M, 32):
                    continue
                variants.append((c, 1))
                for Gw in (256, 384, 512, 768, 1024, 1280, 2048):
                    if Gw > U or U / Gw < 1.5 or Gw < T // 4 or Gw == T:
                        continue
                    variants.append((c, -Gw))
        row = {}
        for cfg, sk in variants:
            def run(W):
                return lib.paella_test_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr() if a.act else None, None, C.data_ptr() if C is not None else None,
                                                 C16.data_ptr() if C16 is not None else None, M, N, K, a.act, stats.data_ptr() if a.ln else None, cfg, sk,
                                                 ws.data_ptr(), ws.numel(), st())
            if run(Ws[0]) != 0:
                continue
            torch.cuda.synchronize()
            ts = []
            for _ in range(3 if big else 6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for W in Ws:
                    run(W)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / ncopy)
            ts.sort()
            row["%d/%d" % (cfg, sk)] = round(ts[len(ts) // 2], 2)
        flops = 2.0 * M * N * K
        top = sorted((v, k) for k, v in row.items() if not k.startswith("-1"))[:6]
        heur = row.get("-1/1")
        results[name] = {"MNK": [M, N, K], "us": row, "best": top[0][1], "best_us": top[0][0], "best_tflops": round(flops / top[0][0] / 1e6, 1),
                         "heuristic_us": heur, "heuristic_tflops": round(flops / heur / 1e6, 1) if heur else None}
        print("%-34s heuristic %9.1f us %7.1f TF | " % (name, heur or -1, flops / (heur or 1e30) / 1e6) + "  ".join("%s %.1f (%.0f TF)" % (k, v, flops / v / 1e6) for v, k in top), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(results, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
