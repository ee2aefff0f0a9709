#!/bin/bash
# Same-box A/B inside ONE tree through a test hook (bench.py --hook NAME=INT), plus the HEAD baseline copy under ab_base/ when present.
# usage (GPU box): bash tools/ab_r06c.sh <tag> <hook> <valA> <valB> ; writes gpurun_out/ab_<tag>.txt
tag=$1; hook=$2; va=$3; vb=$4
out=gpurun_out/ab_$tag.txt
: > $out
run() {  # dir label args...
  d=$1; shift; l=$1; shift
  ( cd $d && python bench.py --no-cpu-baseline --no-extra --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-14s %-64s %9.3f ms/step  %8.3f img/s' % ('$l', ' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))" "$@" ) >> $out
}
for rep in 1 2; do
  for cfg in "--batch 32 --steps 3 --warmup 1" "--batch 128 --steps 2 --warmup 1" "--batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1" "--gemm bf16 --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1"; do
    [ -d ab_base ] && [ $rep = 1 ] && run ab_base base $cfg
    run . "$hook=$va" $cfg --hook $hook=$va
    run . "$hook=$vb" $cfg --hook $hook=$vb
  done
done
[ -d ab_base ] && run ab_base base --batch 1 --steps 20 --warmup 3
run . new --batch 1 --steps 20 --warmup 3
[ -d ab_base ] && run ab_base base --batch 1 --steps 20 --warmup 3
run . new --batch 1 --steps 20 --warmup 3
cat $out
