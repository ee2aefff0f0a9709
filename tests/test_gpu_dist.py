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


# ---------------------------------------------------------------------------------------------------------------------
# The N > 1 code paths of bench.py on a ONE-GPU box: two ranks share cuda:0 and talk over gloo (which moves device tensors through
# the host; RCCL refuses two ranks on one device).  Everything that depends on rank != 0 or world > 1 runs for real: the receive buffer of the packed
# conditioning broadcast, shard rows [B, 2B) keyed by a NON-ZERO device-resident row offset through the captured graphs (GraphSampler / GraphInpainter), the
# all_reduce(MAX) timing, the readiness agreement before each throughput workload, the barriers around rank 0's roofline pass.  `--rehearsal` shrinks every
# workload to the tiny model and adds the check that matters: the gathered shards equal the unsharded request bit for bit, tokens AND images.
# ---------------------------------------------------------------------------------------------------------------------
REHEARSAL = ["--gpus", "2", "--rehearsal", "--dist-backend", "gloo", "--share-device", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]


def test_two_rank_rehearsal_of_the_distributed_bench_on_one_device():
    out = _torchrun(2, [os.path.join(ROOT, "bench.py")] + REHEARSAL, timeout=900)
    print(json.dumps(out))
    c = out["config"]
    assert out["n_gpus"] == 2 and c["world_size_observed"] == 2 and c["collective_backend"].startswith("gloo") and c["share_device"] is True
    assert out["value"] > 0 and c["images_per_step"] == 2 and c["submission"] == "hip-graph replay"
    assert 0 < c["per_rank_ms"]["min"] <= c["per_rank_ms"]["max"] <= out["ms_per_step"] * 1.001
    assert c["broadcast_ms"] > 0 and c["graph_replay_ms"] > 0 and c["broadcast_mbytes"] > 0
    r = out["rehearsal"]
    assert r["tokens_equal_unsharded"] and r["images_equal_unsharded"] and r["row_offsets"] == [0, 256]  # rank 1 sampled rows [1, 2) at a non-zero offset
    tp = out["throughput"]
    assert [(t["model"], t["batch"]) for t in tp] == [("tiny", 4), ("tiny", 3), ("tiny", 2)] and not any("error" in t for t in tp)
    for t in tp:
        assert t["n_gpus"] == 2 and t["images_per_step"] == 2 * t["batch"] and t["submission"] == "hip-graph replay"
        assert t["broadcast_ms"] > 0 and t["sampler_ms"] > 0 and t["images_per_sec"] > 0
        assert t["rehearsal"]["tokens_equal_unsharded"] and t["rehearsal"]["images_equal_unsharded"], t["workload"]
        assert t["rehearsal"]["row_offsets"][1] == t["batch"] * 256
    assert "inpaint" in tp[2]["workload"] or "masked renoise" in tp[2]["workload"]  # the configs[4] recipe went through the same path


def test_injected_setup_failure_on_rank_1_is_an_error_entry_not_a_hang():
    out = _torchrun(2, [os.path.join(ROOT, "bench.py")] + REHEARSAL + ["--inject-setup-failure", "1"], timeout=600)
    assert out["value"] > 0 and out["rehearsal"]["tokens_equal_unsharded"]          # the headline is untouched
    tp = out["throughput"]
    assert len(tp) == 3 and all(t.get("error") == "set-up failed on another rank" for t in tp), tp   # rank 0 learned it through the readiness agreement


def test_rehearsal_line_at_world_size_one():
    """the same flags without a launcher: the non-distributed branch of the same code (no collective at all)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearsal", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["collective_backend"] is None and out["config"]["broadcast_ms"] == 0.0
    assert out["rehearsal"]["tokens_equal_unsharded"] and all(t["rehearsal"]["images_equal_unsharded"] for t in out["throughput"])
