"""Probe: two ranks on ONE device over gloo -- do broadcast / all_reduce / all_gather / barrier accept device tensors here?"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
res = {}
for name, fn in [("broadcast", lambda: dist.broadcast(t, src=0)), ("all_reduce_max", lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX)),
                 ("all_reduce_min_int", lambda: dist.all_reduce(ti, op=dist.ReduceOp.MIN)),
                 ("all_gather", lambda: dist.all_gather([torch.empty_like(t) for _ in range(world)], t)), ("barrier", lambda: dist.barrier())]:
    t = torch.full((1 << 20,), float(rank + 1), device=dev)
    ti = torch.tensor([rank], device=dev)
    try:
        fn(); torch.cuda.synchronize(); res[name] = "ok t[0]=%g ti=%d" % (float(t[0]), int(ti))
    except Exception as e:
        res[name] = "FAIL " + repr(e)[:200]
if rank == 0:
    print(res, flush=True)
dist.destroy_process_group()
