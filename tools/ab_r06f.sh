#!/bin/bash
# Same box: the tree of the evidence commit (ab_base/, rebuilt to the commit named in the first line of the output) against this tree, fp32 and fast mode.
out=gpurun_out/ab_tail_draw_ahead.txt
: > $out
run() {  # dir label args...
  d=$1; shift; l=$1; shift
  ( cd $d && python bench.py --no-cpu-baseline --no-extra --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-6s %-74s %9.3f ms/step  %8.3f img/s' % ('$l', ' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))" "$@" ) >> $out
}
for rep in 1 2; do
  for cfg in "--batch 1 --steps 20 --warmup 3" "--batch 32 --steps 3 --warmup 1" "--batch 128 --steps 2 --warmup 1" "--batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1" "--gemm bf16 --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1"; do
    run ab_base base $cfg
    run . new $cfg
  done
done
cat $out
