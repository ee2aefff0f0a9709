#!/bin/bash
# What a 256x256 ping-pong tile costs beyond its K loop: the same tile count and K on shapes whose A panel / W panel stays cache-resident, and K swept at a fixed tile count.
# usage (GPU box, repo root): bash tools/probes/pp_shapes.sh -> gpurun_out/pp_shapes.txt
out=gpurun_out/pp_shapes.txt
: > $out
for cfg in "32768 5120 1280" "32768 5120 2560" "32768 5120 640" "32768 5120 5120" "256 655360 1280" "655360 256 1280" "2048 81920 1280" "81920 2048 1280" "16384 4096 1280" "16384 4096 5120"; do
  for t in "37 1" "37 -256" "36 1" "36 -256"; do
    python tools/probes/gemm_bf16_one.py $cfg $t 6 0 2>/dev/null | tail -1 >> $out
  done
done
cat $out
