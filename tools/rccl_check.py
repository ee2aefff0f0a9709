"""RCCL evidence that fits a 1-GPU lease: run under torch.distributed.run with ANY number of ranks, one per GPU --
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rccl_check.py
It takes the product's multi-GPU path with a LIVE nccl (= RCCL) process group: one packed conditioning broadcast that also carries the Philox seed
(kept on the device), batch shards keyed by global rows, and checks on every rank that
  (a) sample_sharded(..., layout=...) == this rank's rows of the unsharded sample(noise="philox", seed=s), bit for bit, and
  (b) ONE captured GraphSampler replayed with the broadcast conditioning and shard=(lo, total) gives the same rows.
Rank 0 prints ONE JSON line.  World size 1 exercises every line of the collective path (the broadcast is a real RCCL call with one rank)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import paella_amd
from paella_amd import synth
from paella_amd.dist import broadcast_conditioning, cond_spec_layout, sample_sharded, shard_bounds, shard_inputs

TINY = dict(c_in=32, c_out=32, num_labels=64, c_r=16, patch_size=2, c_cond=64, c_hidden=[32, 64, 64], nhead=[-1, 4, 4], blocks=[1, 2, 1],
            level_config=['CT', 'CTA', 'CTA'], clip_embd=48, byt5_embd=40, clip_seq_len=4, kernel_size=3, dropout=0.1, self_attn=True)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    m = paella_amd.Paella(**TINY)
    synth.randomize_(m, seed=0)
    m = m.to(dev)
    per, H = 2, 16
    total = per * world
    S = 3
    mk = lambda seed: synth.synth_conditioning(total, S, TINY["byt5_embd"], TINY["clip_embd"], seed=seed, device=dev)
    cond, uncond = mk(2), mk(3)  # every rank can build them (seeded) -- only rank 0's copies enter the collective
    kw = dict(steps=3, renoise_steps=2, temperature=(1.0, 0.3), cfg=8.0)
    full = paella_amd.sample(m, cond, (total, H, H), unconditional_inputs=uncond, device=dev, noise="philox", seed=77, **kw)
    lo, hi = shard_bounds(total, rank, world)
    lay = cond_spec_layout(m, total, S_byt5=S, clip=True, n_clip_image=0)
    # (a) the eager sharded path: ONE RCCL broadcast (layout known on every rank), seed inside it and kept on the device
    got = sample_sharded(m, cond if rank == 0 else None, uncond if rank == 0 else None, (total, H, H), src=0, layout=lay, seed=77 if rank == 0 else None, **kw)
    ok_a = bool(torch.equal(got, full[lo:hi]))
    # (b) a captured graph replayed with conditioning that arrived through the live process group
    (c_b, u_b), seed_t = broadcast_conditioning([cond, uncond] if rank == 0 else None, src=0, device=dev, layout=lay, seed=77, with_seed=True, seed_on_device=True)
    gs = paella_amd.GraphSampler(m, shard_inputs(c_b, lo, hi), shard_inputs(u_b, lo, hi), (hi - lo, H, H), device=dev, **kw)
    out = gs(shard_inputs(c_b, lo, hi), shard_inputs(u_b, lo, hi), seed=int(seed_t.item()), shard=(lo, total)).clone()
    ok_b = bool(torch.equal(out, full[lo:hi]))
    # (c) a source-side conditioning that does not match the agreed layout fails on EVERY rank after the gather (nobody hangs in a collective)
    bad = dict(cond, byt5=torch.randn(total, S + 1, TINY["byt5_embd"], device=dev))
    ok_c = False
    try:
        sample_sharded(m, bad if rank == 0 else None, uncond if rank == 0 else None, (total, H, H), src=0, layout=lay, gather=True, seed=77 if rank == 0 else None, **kw)
    except ValueError:
        ok_c = True
    flags = torch.tensor([int(ok_a), int(ok_b), int(ok_c)], device=dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"check": "rccl_shard_path", "collective_backend": dist.get_backend() + " (RCCL)", "world_size": dist.get_world_size(),
                          "device": torch.cuda.get_device_name(dev), "images_total": total, "collectives_per_request": 1,
                          "sample_sharded_equals_unsharded_rows": bool(flags[0].item()), "graph_sampler_shard_equals_unsharded_rows": bool(flags[1].item()),
                          "mismatched_layout_raises_on_every_rank_after_the_gather": bool(flags[2].item())}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if bool(flags.min().item()) else 1)


if __name__ == "__main__":
    main()
