"""Is the CPU baseline bench.py reports (the oracle, `cpu_baseline.kind = "port"`) time-equivalent to the REFERENCE's own code on the same cores?
Runs in the AUTHORING container only (it imports /root/reference, which does not exist on the GPU box): the reference's own `Paella` (src/modules.py) driven
by its own `sample()` (src/utils.py:35, imported with the stubs of oracle/make_golden.py) and the oracle's restatement, same seeded 570M-class weights,
same conditioning, batch 1, 32x32 tokens, 8 steps, CFG 8 -- each timed at the same thread count, best of N runs after a warm-up.
Usage: python tools/cpu_ref_vs_port.py [--threads 8] [--runs 2]     (VERDICT r03 item 8 / SURVEY 8(d) CPU baseline plan)"""
import argparse
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch

from oracle import golden_configs as G
from oracle import make_golden as MG
from oracle import paella_oracle as O
from paella_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ref = MG.import_reference()
    cfg = G.UNET_570M
    model, sd = MG.make_ref_unet(ref["modules"], cfg, G.WEIGHT_SEED)
    cond = synth.synth_conditioning(1, 0, cfg["byt5_embd"], cfg["clip_embd"], seed=2)
    uncond = synth.synth_conditioning(1, 0, cfg["byt5_embd"], cfg["clip_embd"], seed=3)
    grid, L = 32, cfg["num_labels"]
    kw = dict(steps=a.steps, renoise_steps=a.steps - 1, temperature=(1.0, 0.2), cfg=8.0, device="cpu")

    def run_reference():
        torch.manual_seed(0)
        t0 = time.perf_counter()
        with torch.no_grad():
            toks = ref["utils"].sample(model, cond, (1, grid, grid), unconditional_inputs=uncond, **kw)
        return time.perf_counter() - t0, toks

    t_list = [float(v) for v in torch.linspace(1.0, 0.0, a.steps + 1)]
    temps = [float(v) for v in torch.linspace(1.0, 0.2, a.steps)]
    noise = O.replay_torch_noise(0, (1, grid, grid), L, a.steps, a.steps - 1)
    fwd = lambda tk, rr, **i: O.unet_forward(sd, cfg, tk, rr, **i)

    def run_port():
        t0 = time.perf_counter()
        with torch.no_grad():
            toks, _ = O.sample(fwd, L, cond, uncond, (1, grid, grid), steps=a.steps, renoise_steps=a.steps - 1, temperatures=temps, cfgs=[(8.0, -7.0)] * a.steps,
                               t_list=t_list, noise=noise)
        return time.perf_counter() - t0, toks

    run_reference(); run_port()  # warm-up (allocator, thread pool)
    tr = [run_reference() for _ in range(a.runs)]
    tp = [run_port() for _ in range(a.runs)]
    br, bp = min(t for t, _ in tr), min(t for t, _ in tp)
    same = bool(torch.equal(tr[0][1], tp[0][1]))
    print("# reference src/utils.py sample() + src/modules.py Paella vs oracle/paella_oracle.py, 570M-class stand-in, batch 1, 32x32 tokens, %d steps, CFG 8, fp32, torch %s"
          % (a.steps, torch.__version__))
    print("# host: %d threads of %d logical CPUs (authoring container)" % (a.threads, os.cpu_count()))
    print("reference: %s s   (best %.2f s)" % (" ".join("%.2f" % t for t, _ in tr), br))
    print("oracle   : %s s   (best %.2f s)" % (" ".join("%.2f" % t for t, _ in tp), bp))
    print("oracle / reference time = %.3f   tokens identical under the same seed: %s" % (bp / br, same))


if __name__ == "__main__":
    main()
