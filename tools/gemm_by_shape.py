"""Per-shape GEMM time INSIDE the model: one eager image of the headline workload with every dense-contraction launch bracketed by HIP events
(paella_prof_detail), grouped by (M, N, K, prologue).  Usage (GPU box): python tools/gemm_by_shape.py [--batch B --grid G --sample-steps S]"""
import argparse
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import paella_amd
from paella_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--grid", type=int, default=32)
ap.add_argument("--sample-steps", type=int, default=8)
ap.add_argument("--gemm", default="fp32", choices=["fp32", "bf16"], help="bf16 = the opt-in fast mode of both models")
ap.add_argument("--model", default="570m", choices=sorted(bench.MODELS))
ap.add_argument("--s-byt5", type=int, default=0)
ap.add_argument("--clip-image", type=int, default=0)
ap.add_argument("--hook", action="append", default=[], metavar="NAME=INT", help="A/B: paella_test_NAME(INT) before the run")
a = ap.parse_args()
lib = _lib.load()
for hk in a.hook:
    _lib.check(getattr(lib, "paella_test_" + hk.split("=")[0])(int(hk.split("=")[1])))
dev = torch.device("cuda")
mcfg, vcfg = bench.MODELS[a.model], bench.VQ[a.model]
model = paella_amd.Paella(**mcfg)
synth.randomize_(model, seed=0)
model = model.to(dev)
vq = paella_amd.VQModel(**vcfg)
synth.randomize_(vq, seed=0)
vq = vq.to(dev)
model.set_gemm_precision(a.gemm)
vq.set_gemm_precision(a.gemm)
mk = lambda n, seed: synth.synth_conditioning(n, a.s_byt5, mcfg["byt5_embd"], mcfg["clip_embd"], seed=seed, n_clip_image=a.clip_image, device=dev)
c, u = mk(a.batch, 2), mk(a.batch, 3)
kw = dict(steps=a.sample_steps, renoise_steps=a.sample_steps - 1, temperature=(1.0, 0.2), cfg=8.0, device=dev, noise="philox")


def once(seed):
    toks = paella_amd.sample(model, c, (a.batch, a.grid, a.grid), unconditional_inputs=u, seed=seed, **kw)
    return vq.decode_indices(toks)


once(1)
once(2)
torch.cuda.synchronize()
lib.paella_prof_enable(1)
once(3)
torch.cuda.synchronize()
cap = 1 << 16
us = np.zeros(cap, dtype=np.float32)
shp = np.zeros(cap * 5, dtype=np.int32)
n = lib.paella_prof_detail(us.ctypes.data_as(ctypes.c_void_p), shp.ctypes.data_as(ctypes.c_void_p), cap)
lib.paella_prof_enable(0)
shp = shp[:n * 5].reshape(n, 5)
groups = collections.OrderedDict()
for t, s in zip(us[:n], shp):
    groups.setdefault(tuple(int(v) for v in s), []).append(float(t))
names = {0: "plain", 1: "GRN", 2: "LN", 3: "conv", 4: "GRNraw"}
print("# python tools/gemm_by_shape.py --gemm %s %s --model %s --s-byt5 %d --clip-image %d --batch %d --grid %d --sample-steps %d: %d GEMM launches per image batch, event-timed, eager; per (M, N, K, prologue)"
      % (a.gemm, " ".join("--hook " + h for h in a.hook), a.model, a.s_byt5, a.clip_image, a.batch, a.grid, a.sample_steps, n))
print("%-8s %-7s %-7s %-6s %6s %9s %9s %9s %7s" % ("M", "N", "K", "pro", "calls", "avg us", "min us", "total ms", "TF/s"))
tot = 0.0
for (M, N, K, pro, tail), ts in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    tt = sum(ts)
    tot += tt
    print("%-8d %-7d %-7d %-6s %6d %9.2f %9.2f %9.3f %7.1f" % (M, N, K, names[pro] + ("+tail" if tail else ""), len(ts), tt / len(ts), min(ts), tt / 1e3, 2.0 * M * N * K * len(ts) / tt / 1e6))
print("total %.3f ms over %d launches" % (tot / 1e3, n))
