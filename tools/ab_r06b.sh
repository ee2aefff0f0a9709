#!/bin/bash
# Same-box A/B of the tree against a HEAD baseline copy under ab_base/ (built in the authoring container; not tracked).
# usage (GPU box): bash tools/ab_r06b.sh <tag> ; writes gpurun_out/ab_<tag>.txt
tag=${1:-x}
out=gpurun_out/ab_$tag.txt
: > $out
run() {  # dir label args...
  d=$1; shift; l=$1; shift
  ( cd $d && python bench.py --no-cpu-baseline --no-extra --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-10s %-44s %9.3f ms/step  %8.3f img/s' % ('$l', ' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))" "$@" ) >> $out
}
for rep in 1 2; do
  for cfg in "--batch 1 --steps 20 --warmup 3" "--batch 32 --steps 3 --warmup 1" "--batch 128 --steps 2 --warmup 1" "--batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1"; do
    run ab_base base $cfg
    run . new $cfg
  done
done
for cfg in "--gemm bf16 --batch 32 --steps 3 --warmup 1" "--gemm bf16 --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1"; do
  run ab_base base $cfg
  run . new $cfg
done
cat $out
