"""GPU: the multi-GPU path with a LIVE RCCL process group on whatever GPUs the box has (a 1-GPU lease runs it at world size 1).

tests/test_dist.py covers the host logic on gloo / CPU with world size 2; this file makes sure the code the driver's --gpus N scaling run executes --
torch.distributed over nccl (= RCCL), the packed conditioning broadcast with the seed inside, shard-exact Philox through a captured graph -- has
run against RCCL at least once (reference: the reference has no sampling collective, src_distributed/utils.py:85-94 is DDP setup)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(nproc, script_args, timeout=600):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, "torchrun failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_rccl_shard_path_equals_unsharded():
    n = max(1, torch.cuda.device_count())
    out = _torchrun(n, [os.path.join(ROOT, "tools", "rccl_check.py")])
    print(json.dumps(out))
    assert out["collective_backend"] == "nccl (RCCL)" and out["world_size"] == n
    assert out["sample_sharded_equals_unsharded_rows"] and out["graph_sampler_shard_equals_unsharded_rows"]
    assert out["mismatched_layout_raises_on_every_rank_after_the_gather"]


def test_bench_runs_through_rccl_at_the_world_size_of_the_box():
    """bench.py under the launcher the driver uses, forced onto the distributed path even at one rank: the line must report the RCCL backend."""
    n = max(1, torch.cuda.device_count())
    out = _torchrun(n, [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--force-dist", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"])
    print(json.dumps({k: out[k] for k in ("value", "n_gpus", "ms_per_step", "config")}))
    assert out["config"]["collective_backend"] == "nccl (RCCL)" and out["config"]["world_size_observed"] == n and out["n_gpus"] == n
    assert out["value"] > 0 and out["scaling"] == "weak"
    # attribution of a step, so that a sub-linear point of a scaling run can be explained from the record alone: wall time per rank (min / max over ranks),
    # event-timed conditioning broadcast + shard slicing, event-timed sampler
    c = out["config"]
    assert set(c["per_rank_ms"]) == {"min", "max"} and 0 < c["per_rank_ms"]["min"] <= c["per_rank_ms"]["max"] <= out["ms_per_step"] * 1.001
    assert c["broadcast_ms"] > 0 and c["graph_replay_ms"] > 0 and c["broadcast_ms"] + c["graph_replay_ms"] <= out["ms_per_step"] * 1.05
