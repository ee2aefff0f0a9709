"""CPU, world_size 2, gloo: the multi-GPU host logic (one conditioning broadcast + batch sharding)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from paella_amd.dist import broadcast_conditioning, conditioning_layout, shard_bounds, shard_inputs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        B = 5
        cond = {"byt5": torch.randn(B, 3, 8, generator=g), "clip": torch.randn(B, 6, generator=g),
                "clip_image": [torch.randn(B, 6, generator=g), torch.randn(B, 6, generator=g)]}
        uncond = {"byt5": torch.randn(B, 1, 8, generator=g), "clip": torch.randn(B, 6, generator=g), "clip_image": None}
        got_c, got_u = broadcast_conditioning([cond, uncond] if rank == 0 else None, src=0, device="cpu")
        ok = torch.equal(got_c["byt5"], cond["byt5"]) and torch.equal(got_c["clip"], cond["clip"])
        ok = ok and all(torch.equal(a, b) for a, b in zip(got_c["clip_image"], cond["clip_image"]))
        ok = ok and torch.equal(got_u["byt5"], uncond["byt5"]) and got_u["clip_image"] is None
        # fixed-shape serving: every rank derives the layout from same-shaped tensors -> one collective, no shape exchange
        layout = conditioning_layout([cond, uncond])  # (both ranks built the same tensors from the same seed here)
        lay_c, lay_u = broadcast_conditioning([cond, uncond] if rank == 0 else None, src=0, device="cpu", layout=layout)
        ok = ok and torch.equal(lay_c["byt5"], cond["byt5"]) and torch.equal(lay_u["clip"], uncond["clip"]) and lay_u["clip_image"] is None
        ok = ok and all(torch.equal(a, b) for a, b in zip(lay_c["clip_image"], cond["clip_image"]))
        lo, hi = shard_bounds(B, rank, world)
        mine = shard_inputs(got_c, lo, hi)
        # gather the shards back: sharded == unsharded
        parts = [None] * world
        dist.all_gather_object(parts, (lo, hi, mine["byt5"]))
        rebuilt = torch.cat([p[2] for p in sorted(parts, key=lambda p: p[0])], dim=0)
        ok = ok and torch.equal(rebuilt, cond["byt5"])
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_shard_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    spans = sorted(s for _, _, s in res)
    assert spans == [(0, 3), (3, 5)]
