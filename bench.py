#!/usr/bin/env python
"""Headline benchmark: images/sec + single-image ms for Paella sampling, 256x256 px (32x32 tokens) @ 8 steps.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
One "step" = one pass of the hot path over one batch of synthetic input:
    sample() -- 8 denoising steps, classifier-free guidance 8.0 (cond + uncond rows batched per evaluation),
    temperature 1.0 -> 0.2, renoise 7 -- followed by VQGAN f8 decode_indices to 256x256 px.
Workload = BASELINE.json configs[1]: the 573M-class denoiser (stand-in Paella(blocks=[4,8,4]) = 570.3M params),
32x32 token grid, CLIP-text-only conditioning (byt5 of length 0), batch 1 per GPU, fp32 end to end,
seeded synthetic weights and random embeddings, inputs resident in HBM when the timed region starts.
With N GPUs every rank generates its own batch (weak scaling); rank 0 owns the conditioning of all N*batch images and
broadcasts it once per step over RCCL (the only collective of the path).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MODELS = {
    "570m": dict(c_in=256, c_out=256, num_labels=8192, c_r=64, patch_size=2, c_cond=1024, c_hidden=[640, 1280, 1280],
                 nhead=[-1, 16, 16], blocks=[4, 8, 4], level_config=['CT', 'CTA', 'CTA'], clip_embd=1024, byt5_embd=1536,
                 clip_seq_len=4, kernel_size=3, dropout=0.1, self_attn=True),
    "tiny": dict(c_in=32, c_out=32, num_labels=64, c_r=16, patch_size=2, c_cond=64, c_hidden=[32, 64, 64], nhead=[-1, 4, 4],
                 blocks=[1, 2, 1], level_config=['CT', 'CTA', 'CTA'], clip_embd=48, byt5_embd=40, clip_seq_len=4, kernel_size=3,
                 dropout=0.1, self_attn=True),
}
MODELS["1b"] = dict(MODELS["570m"], blocks=[6, 16, 6])
VQ = {"570m": dict(levels=3, bottleneck_blocks=12, c_hidden=384, c_latent=4, codebook_size=8192, scale_factor=0.3764),
      "tiny": dict(levels=3, bottleneck_blocks=2, c_hidden=64, c_latent=4, codebook_size=64, scale_factor=0.3764)}
VQ["1b"] = VQ["570m"]
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense
# SURVEY.md section 8(d), per image: 2 * steps * F_fwd(model, grid, S=4) + VQGAN f8 decode, in GFLOP (the GEMM-shaped work)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (v_mfma_f32_16x16x32_bf16), MI355X_MICROARCH.md
WORKLOAD_TAG = {("570m", 1, 32, 8): "BASELINE configs[1]", ("570m", 64, 64, 12): "BASELINE configs[2]"}
ALGO_GFLOP_PER_IMAGE = {("570m", 32, 8): 2 * 8 * 66.27 + 38.8}
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (BASELINE configs[1]: 1)")
    ap.add_argument("--model", default="570m", choices=sorted(MODELS))
    ap.add_argument("--grid", type=int, default=32, help="token grid side (32 = 256 px at f8)")
    ap.add_argument("--sample-steps", type=int, default=8)
    ap.add_argument("--noise", default="philox", choices=["philox", "torch"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured HIP graph per image batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational batched-throughput run")
    ap.add_argument("--extra-batch", type=int, default=8)
    ap.add_argument("--gemm", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = the exact path (the headline); bf16 = OPT-IN fast mode outside the parity contract (bf16 MFMA operands, fp32 accumulate)")
    ap.add_argument("--force-dist", action="store_true", help="take the torch.distributed (RCCL) path even at world size 1 (launch under torchrun)")
    return ap.parse_args()


def gen_images(model, vq, cond, uncond, batch, grid, sample_steps, noise, seed, device):
    import paella_amd
    toks = paella_amd.sample(model, cond, (batch, grid, grid), unconditional_inputs=uncond, steps=sample_steps,
                             renoise_steps=sample_steps - 1, temperature=(1.0, 0.2), cfg=8.0, device=device, noise=noise, seed=seed)
    return vq.decode_indices(toks)


def timed(fn, steps, warmup, distributed, device):
    for _ in range(warmup):
        fn()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(model_cfg, vq_cfg, grid, sample_steps, unet_sd, vq_sd, cond, uncond):
    """The oracle (CPU restatement of the reference, oracle/paella_oracle.py) timed on this host's cores, on ONE image of
    the same workload: sample_steps x 2 forwards + sampling tails + f8 decode."""
    from oracle import paella_oracle as O
    n = torch.get_num_threads()
    L = model_cfg["num_labels"]
    noise = O.replay_torch_noise(0, (1, grid, grid), L, sample_steps, sample_steps - 1)
    t_list = [float(v) for v in torch.linspace(1.0, 0.0, sample_steps + 1)]
    temps = [float(v) for v in torch.linspace(1.0, 0.2, sample_steps)]
    cf = (8.0, -7.0)
    fwd = lambda tk, rr, **i: O.unet_forward(unet_sd, model_cfg, tk, rr, **i)
    c1 = {k: (v[:1].cpu() if v is not None else None) for k, v in cond.items()}
    u1 = {k: (v[:1].cpu() if v is not None else None) for k, v in uncond.items()}
    with torch.no_grad():
        t0 = time.perf_counter()
        toks, _ = O.sample(fwd, L, c1, u1, (1, grid, grid), steps=sample_steps, renoise_steps=sample_steps - 1, temperatures=temps,
                           cfgs=[cf] * sample_steps, t_list=t_list, noise=noise)
        O.vq_decode_indices(vq_sd, vq_cfg, toks % vq_cfg["codebook_size"])
        dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "images/sec", "cores": n, "kind": "port",
            "sample": "1 image: %d steps x 2 forwards (cond+uncond) + sampling tails + VQGAN f8 decode, torch CPU fp32, %.1f s" % (sample_steps, dt)}


def main():
    a = parse()
    distributed = a.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or (a.force_dist and "RANK" in os.environ)
    rank, world, local = 0, 1, 0
    if distributed:
        rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    import paella_amd
    from paella_amd import _lib, synth
    from paella_amd.dist import broadcast_conditioning, conditioning_layout, shard_bounds, shard_inputs
    lib = _lib.load()  # fails loudly if the HIP library is missing
    paella_amd.set_gemm_precision(a.gemm)

    mcfg, vcfg = MODELS[a.model], VQ[a.model]
    model = paella_amd.Paella(**mcfg)
    unet_sd = synth.randomize_(model, seed=0)
    model = model.to(device)
    vq = paella_amd.VQModel(**vcfg)
    vq_sd = synth.randomize_(vq, seed=0)
    vq = vq.to(device)

    total = a.batch * world
    # rank 0 owns the conditioning of the whole job (as if it had run the CLIP text encoder); CLIP-text-only: S_byt5 = 0
    cond_all = uncond_all = None
    if rank == 0:
        cond_all = synth.synth_conditioning(total, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=2, device=device)
        uncond_all = synth.synth_conditioning(total, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=3, device=device)
    lo, hi = shard_bounds(total, rank, world)
    # fixed request shapes: every rank derives the broadcast layout locally, so the per-step exchange is ONE async RCCL broadcast
    tmpl = synth.synth_conditioning(total, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=2, device=device)
    layout = conditioning_layout([tmpl, tmpl]) if distributed else None
    counter = [0]
    use_graph = not a.no_graph and a.noise == "philox"
    sampler = None
    if use_graph:
        # capture sample() + decode once for this rank's shapes; every step replays it with fresh conditioning / seed
        c0 = synth.synth_conditioning(a.batch, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=2, device=device)
        u0 = synth.synth_conditioning(a.batch, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=3, device=device)
        sampler = paella_amd.GraphSampler(model, c0, u0, (a.batch, a.grid, a.grid), steps=a.sample_steps, renoise_steps=a.sample_steps - 1,
                                          temperature=(1.0, 0.2), cfg=8.0, device=device, vqgan=vq)

    def step():
        if distributed:
            c, u = broadcast_conditioning([cond_all, uncond_all] if rank == 0 else None, src=0, device=device, layout=layout)
            c, u = shard_inputs(c, lo, hi), shard_inputs(u, lo, hi)
        else:
            c, u = cond_all, uncond_all
        counter[0] += 1
        if sampler is not None:
            return sampler(c, u, seed=1000 * counter[0] + rank)[1]
        return gen_images(model, vq, c, u, a.batch, a.grid, a.sample_steps, a.noise, 1000 * counter[0] + rank, device)

    def step_eager():
        counter[0] += 1
        c, u = (cond_all, uncond_all) if not distributed else (shard_inputs(cond_all, lo, hi), shard_inputs(uncond_all, lo, hi))
        return gen_images(model, vq, c, u, a.batch, a.grid, a.sample_steps, a.noise, 1000 * counter[0] + rank, device)

    dt = timed(step, a.steps, a.warmup, distributed, device)
    images = total * a.steps
    value = images / dt
    ms_per_step = dt / a.steps * 1e3

    # ---- roofline of the dominant kernel family (fp32 MFMA GEMM): one extra identical pass with every GEMM launch
    # bracketed by HIP events on its stream (eager launches; the timed region above runs without the events)
    roof = None
    if rank == 0:
        lib.paella_prof_enable(1)
        step_eager()
        torch.cuda.synchronize(device)
        ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.paella_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
        lib.paella_prof_enable(0)
        ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # HBM traffic per launch from the committed PMC pass of this same workload (bench.py cannot collect PMC counters itself)
        traffic, traffic_note = None, None
        tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            w = tj.get("workload", {})
            if (w.get("model"), w.get("batch_per_gpu"), w.get("grid"), w.get("sample_steps")) == (a.model, a.batch, a.grid, a.sample_steps):
                traffic = tj["hbm_bytes_per_launch"]
                traffic_note = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/r01_pmc_traffic.json): 2*FETCH+WRITE KiB per launch, gfx950 correction applied"
        # algorithmic work of one image as the reference executes it = 2 x sample_steps full forwards
        # + one VQGAN decode.  The CFG de-duplication executes fewer FLOPs for the same result, so both
        # figures are reported: `achieved` prices the algorithmic work, `executed_tflops` what the launches really multiplied.
        algo = ALGO_GFLOP_PER_IMAGE.get((a.model, a.grid, a.sample_steps))
        peak = PEAK_FP32_MFMA_TFLOPS if a.gemm == "fp32" else PEAK_BF16_MFMA_TFLOPS
        executed = ach
        if algo is not None and ms.value > 0:
            ach = algo * a.batch * 1e9 / (ms.value * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": ("gemm_nt_kernel (fp32 v_mfma_f32_16x16x4_f32, all tile configs, split-K reduce included)" if a.gemm == "fp32"
                                            else "gemm_bf16_kernel (v_mfma_f32_16x16x32_bf16) + fp32 gemm_nt_kernel for the few GEMMs without a bf16 path"),
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "achieved_basis": ("SURVEY 8(d) algorithmic GFLOP per image (2 x steps full forwards + decode) / summed GEMM launch time"
                                   if algo is not None else "executed GEMM FLOPs / summed GEMM launch time"),
                "executed_tflops": round(executed, 2), "executed_frac": round(executed / peak, 4),
                "algorithmic_gflop_per_image": algo,
                "traffic": traffic, "traffic_source": traffic_note, "algorithmic_bytes_per_launch": round(by.value / max(n.value, 1)),
                "launches_per_step": int(n.value), "avg_launch_us": round(ms.value * 1e3 / max(n.value, 1), 2),
                "gemm_ms_per_step": round(ms.value, 3), "executed_gflop_per_step": round(fl.value / 1e9, 1),
                "algorithmic_gbytes_per_step": round(by.value / 1e9, 2),
                "hbm_equiv_gbs": round(by.value / (ms.value * 1e-3) / 1e9, 1) if ms.value > 0 else 0.0}
    elif distributed:
        pass
    if distributed:
        dist.barrier()

    extra = None
    if rank == 0 and not a.no_extra and not distributed and a.extra_batch > a.batch:
        eb = a.extra_batch
        ce = synth.synth_conditioning(eb, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=2, device=device)
        ue = synth.synth_conditioning(eb, 0, mcfg["byt5_embd"], mcfg["clip_embd"], seed=3, device=device)
        if use_graph:
            gs = paella_amd.GraphSampler(model, ce, ue, (eb, a.grid, a.grid), steps=a.sample_steps, renoise_steps=a.sample_steps - 1,
                                         temperature=(1.0, 0.2), cfg=8.0, device=device, vqgan=vq)
            fn = lambda: gs(ce, ue, seed=77)
        else:
            fn = lambda: gen_images(model, vq, ce, ue, eb, a.grid, a.sample_steps, a.noise, 77, device)
        k = max(2, a.steps // 3)
        dte = timed(fn, k, 1, False, device)
        extra = {"batch": eb, "images_per_sec": round(eb * k / dte, 3), "ms_per_image": round(dte / (eb * k) * 1e3, 3)}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:  # the CPU baseline is an N = 1 figure
        try:
            cpu = cpu_baseline(mcfg, vcfg, a.grid, a.sample_steps, unet_sd, vq_sd, cond_all, uncond_all)
        except Exception as e:  # the baseline is informational; never lose the GPU line over it
            cpu = {"value": None, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port", "sample": "failed: %r" % (e,)}

    if rank == 0:
        n_params = sum(p.numel() for p in model.parameters())
        line = {
            "metric": "images/sec (whole node) + single-image ms, 256x256 @ 8 steps", "value": round(value, 4), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "single_image_ms": round(ms_per_step / a.batch, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.gemm == "fp32" else "bf16 MFMA operands / f32 accumulate and f32 everywhere else (opt-in fast mode, outside the parity contract)", "data": "synthetic (seeded random-init weights, random CLIP-text embeddings)",
            "config": {"workload": WORKLOAD_TAG.get((a.model, a.batch, a.grid, a.sample_steps), "custom") + ": Paella 573M-class (stand-in blocks=[4,8,4], %.1fM params), %dx%d tokens = %d px, "
                                   "%d steps, CFG 8.0, CLIP-H-text only (S=4), batch %d per GPU, + VQGAN f8 decode"
                                   % (n_params / 1e6, a.grid, a.grid, a.grid * 8, a.sample_steps, a.batch),
                       "model": a.model, "batch_per_gpu": a.batch, "global_batch": total, "grid": a.grid, "sample_steps": a.sample_steps,
                       "noise": a.noise, "submission": "hip-graph replay" if use_graph else "eager launches", "parallelism": "batch-shard x%d, one conditioning broadcast per step" % world},
            "roofline": roof, "cpu_baseline": cpu, "batched": extra,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
