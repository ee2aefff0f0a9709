#!/bin/bash
# SQ / TCC counters of the 256x256 ping-pong tile (37) and the 256x128 tile (36) on one short-K and one long-K shape, no output stored (main loop + ramp only).
# usage (GPU box, repo root): bash tools/probes/pp_counters.sh  -> gpurun_out/pp_counters.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pp_counters
mkdir -p $O
out=$R/gpurun_out/pp_counters.txt
: > $out
cd /tmp && export TMPDIR=/tmp
db() { find $1 -name "*.db" | head -1; }
one() {  # label M N K tile sk
  local l=$1; shift
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    rm -rf $O/t
    rocprofv3 --kernel-trace --pmc $pass --output-format rocpd -d $O/t -- python $R/tools/probes/gemm_bf16_one.py "$@" 6 0 > $O/log.txt 2>&1
    { echo "## $l  ($*)  counters: $pass"; python $R/tools/pmc_summary.py $(db $O/t) gemm_nt_kernel; } >> $out 2>&1
  done
}
one "pingpong K=1280, one tile per workgroup" 32768 5120 1280 37 1
one "pingpong K=1280, 256 persistent ranges" 32768 5120 1280 37 -256
one "pingpong K=5120, one tile per workgroup" 32768 1280 5120 37 1
one "256x128  K=1280, 256 persistent ranges" 32768 5120 1280 36 -256
one "256x128  K=5120, one tile per workgroup" 32768 1280 5120 36 1
cat $out
