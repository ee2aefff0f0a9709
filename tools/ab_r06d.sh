#!/bin/bash
# Fast-mode A/B of the long-K launch rule (256x256 ping-pong tile) through the test hook, plus the HEAD baseline copy; then the full GPU suite.
out=gpurun_out/ab_bf16_pp_rule.txt
: > $out
run() {  # dir label args...
  d=$1; shift; l=$1; shift
  ( cd $d && python bench.py --no-cpu-baseline --no-extra --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-18s %-84s %9.3f ms/step  %8.3f img/s' % ('$l', ' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))" "$@" ) >> $out
}
for rep in 1 2; do
  for cfg in "--gemm bf16 --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1" "--gemm bf16 --batch 128 --steps 2 --warmup 1" "--gemm bf16 --batch 32 --steps 3 --warmup 1"; do
    [ -d ab_base ] && [ $rep = 1 ] && run ab_base base $cfg
    run . "no-pingpong" $cfg --hook gemm_bf16_rule=4
    run . "pingpong" $cfg --hook gemm_bf16_rule=0
  done
done
cat $out
