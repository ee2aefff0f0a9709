"""Per-launch-site (tile, workgroup count) sweep of the MID-SIZE GEMM class inside the model: every distinct site of one batched image batch
whose work lies between the skinny ring class and the one-tile-per-workgroup regime is re-run with each candidate installed through the test hook
(paella_test_gemm_site_cfg) and event-timed IN the eager model (paella_prof_detail: the launch sits between its real neighbours, operands as cold / warm as in the
product); the winners are then confirmed on graph replays of the whole batch.  Prints a table and the C initialiser for gemm.hip's g_sites.
Usage (GPU box): python tools/site_tune_mid.py [--batch 32] [--reps 3]"""
import argparse
import collections
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import paella_amd
from paella_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=3, help="eager image batches per candidate")
ap.add_argument("--min-macs", type=float, default=1.2e9)
ap.add_argument("--max-t128", type=int, default=1023, help="sites with fewer 128x128 tiles than this (above: one tile per workgroup, not swept)")
ap.add_argument("--apply", action="append", default=[], metavar="M,N,K,apro,cfg,G", help="install these entries before the baseline (confirming an earlier sweep)")
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda")
mcfg, vcfg = bench.MODELS["570m"], bench.VQ["570m"]
model = paella_amd.Paella(**mcfg)
synth.randomize_(model, seed=0)
model = model.to(dev)
vq = paella_amd.VQModel(**vcfg)
synth.randomize_(vq, seed=0)
vq = vq.to(dev)
mk = lambda n, seed: synth.synth_conditioning(n, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=seed, device=dev)
B = a.batch
c, u = mk(B, 2), mk(B, 3)
kw = dict(steps=8, renoise_steps=7, temperature=(1.0, 0.2), cfg=8.0, device=dev)
TILES = {0: (128, 128), 10: (128, 128), 18: (64, 64), 34: (64, 64), 30: (32, 32), 31: (32, 32), 32: (32, 64), 33: (64, 32), 35: (32, 64)}
for e in a.apply:
    M, N, K, apro, cfg, G = (int(v) for v in e.split(","))
    _lib.check(lib.paella_test_gemm_site_cfg(M, N, K, apro, 0, cfg, G))


def eager(seed):
    return vq.decode_indices(paella_amd.sample(model, c, (B, 32, 32), unconditional_inputs=u, noise="philox", seed=seed, **kw))


def detail(reps):
    """{(M, N, K, apro): mean us of the launches of that site} over `reps` eager image batches"""
    acc = collections.defaultdict(list)
    for r in range(reps):
        lib.paella_prof_enable(1)
        eager(3 + r)
        torch.cuda.synchronize()
        cap = 1 << 16
        us = np.zeros(cap, dtype=np.float32)
        shp = np.zeros(cap * 5, dtype=np.int32)
        n = lib.paella_prof_detail(us.ctypes.data_as(ctypes.c_void_p), shp.ctypes.data_as(ctypes.c_void_p), cap)
        lib.paella_prof_enable(0)
        for t, s5 in zip(us[:n], shp[:n * 5].reshape(n, 5)):
            M, N, K, pro, tail = (int(v) for v in s5)
            if tail or pro == 3:
                continue
            acc[(M, N, K, 1 if pro in (1, 4) else (2 if pro == 2 else 0))].append(float(t))
    return {k: (sum(v) / len(v), len(v) // reps) for k, v in acc.items()}


def graph_ms(reps=5):
    gs = paella_amd.GraphSampler(model, c, u, (B, 32, 32), vqgan=vq, **kw)
    gs(seed=5)
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        t0 = time.perf_counter()
        gs(seed=7 + r)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


eager(1); eager(2)
base = detail(a.reps)
g0 = graph_ms()
sites = []
for (M, N, K, apro), (t, calls) in base.items():
    macs = float(M) * N * K
    t128 = -(-M // 128) * -(-N // 128)
    if macs >= a.min_macs and t128 <= a.max_t128 and K % 32 == 0 and M >= 256:
        sites.append(((M, N, K, apro), t, calls))
sites.sort(key=lambda s: -s[1] * s[2])
print("# python tools/site_tune_mid.py --batch %d: %d mid-size sites of one image batch (570M, 32x32 tokens, 8 steps); graph replay with the global rules %.2f ms per batch = %.3f ms per image"
      % (B, len(sites), g0, g0 / B), flush=True)
print("# per site: event-timed mean us of its launches inside the eager model, by (tile:G); tile ids: 10 = 128x128 8 waves, 18 = 64x64 direct-to-LDS, 34 = 64x64 ring, 30 / 31 = 32x32 ring")
best = {}
for (M, N, K, apro), t0, calls in sites:
    ktiles = K // 32
    res = {}
    for cfg in (10, 18, 34, 31 if apro == 2 else 30):
        bm, bn = TILES[cfg]
        T = -(-M // bm) * -(-N // bn)
        U = T * ktiles
        cands = sorted(set(g for g in (T, 256, 512, 768, 1024, 1280, 2 * T) if min(T, 256) <= g <= U and (apro != 2 or g % T == 0 or g == T)))
        if cfg in (30, 31) and T > 4096:
            continue
        for g in cands:
            if lib.paella_test_gemm_site_cfg(M, N, K, apro, 0, cfg, g) != 0:
                continue
            try:
                d = detail(a.reps)
                res[(cfg, g)] = d[(M, N, K, apro)][0]
            except Exception as e:  # a candidate the launcher refuses (workspace, prologue without that tile)
                res[(cfg, g)] = float("inf")
    lib.paella_test_gemm_site_cfg(M, N, K, apro, 0, -1, 0)  # remove the run-time entry
    again = detail(a.reps)[(M, N, K, apro)][0]
    kb = min(res, key=res.get)
    keep = res[kb] < min(t0, again) * 0.97
    print("site %5dx%5dx%5d pro %d (%3d launches): rule %.1f / %.1f us (%.1f TF) | " % (M, N, K, apro, calls, t0, again, 2.0 * M * N * K / min(t0, again) / 1e6) +
          " ".join("%d:%d=%.1f" % (k[0], k[1], v) for k, v in sorted(res.items(), key=lambda kv: kv[1])[:6]) +
          ("  -> tile %d G %d (%.1f TF, -%.1f us x %d)" % (kb[0], kb[1], 2.0 * M * N * K / res[kb] / 1e6, min(t0, again) - res[kb], calls) if keep else "  -> kept"), flush=True)
    if keep:
        best[(M, N, K, apro)] = kb
for (M, N, K, apro), (cfg, g) in best.items():
    lib.paella_test_gemm_site_cfg(M, N, K, apro, 0, cfg, g)
g1 = graph_ms()
print("# graph replay with the %d site entries installed: %.2f ms per batch = %.3f ms per image (%.2f %%)" % (len(best), g1, g1 / B, (g1 / g0 - 1) * 100))
print("# g_sites initialiser (M, N, K, prologue class, bf16, G, tile id + 1):")
for (M, N, K, apro), (cfg, g) in best.items():
    print("    {%d, %d, %d, %d, 0, %d, %d}," % (M, N, K, apro, g, cfg + 1))
