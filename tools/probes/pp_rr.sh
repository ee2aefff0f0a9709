#!/bin/bash
# Round-robin whole-tile walk (256 persistent workgroups) against one tile per workgroup and contiguous persistent ranges, tiles 36 / 37, no store and fp32 store.
out=gpurun_out/pp_rr.txt
: > $out
RR=-16777472   # -( (1 << 24) + 256 )
for cfg in "32768 5120 1280" "32768 3840 1280" "32768 1280 1280" "32768 1280 5120" "131072 2560 640" "8192 5120 1280"; do
  for st in 0 1; do
    for t in "37 1" "37 -256" "37 $RR" "36 1" "36 -256" "36 $RR"; do
      python tools/probes/gemm_bf16_one.py $cfg $t 6 $st 2>/dev/null | tail -1 >> $out
    done
  done
done
cat $out
