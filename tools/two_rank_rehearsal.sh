#!/bin/bash
# The N > 1 code paths of bench.py on a ONE-GPU box: two ranks on cuda:0 over gloo, tiny model (tests/test_gpu_dist.py runs the same command and asserts on it).
# Usage (GPU box, repo root): bash tools/two_rank_rehearsal.sh r06   -> gpurun_out/r06_two_rank_rehearsal.txt
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
CMD="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 2 --rehearsal --dist-backend gloo --share-device --steps 2 --warmup 1 --no-cpu-baseline"
{ echo "# $CMD   (MI355X, ONE device shared by two ranks; TEST ONLY -- exercises every rank != 0 / world > 1 branch, not a measurement)"
  $CMD 2> $O/log_two_rank_rehearsal.txt | grep '^{'
  echo "# the same with --inject-setup-failure 1: every throughput workload must come back as an error entry on rank 0, no hang"
  $CMD --inject-setup-failure 1 2>> $O/log_two_rank_rehearsal.txt | grep '^{'; } > $O/${TAG}_two_rank_rehearsal.txt
tail -c 1500 $O/${TAG}_two_rank_rehearsal.txt
