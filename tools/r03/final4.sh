#!/bin/bash
# refresh the batch-32 and configs[2] kernel traces on the final kernels
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_final4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra --no-graph"
db() { find $1 -name "*.db" | head -1; }
trace() {  # name, images, command...
    local name=$1 images=$2; shift 2
    rocprofv3 --kernel-trace --output-format rocpd -d $O/tmp_$name -- "$@" > $O/log_trace_$name.txt 2>&1
    { echo "# rocprofv3 --kernel-trace --output-format rocpd -- $* ; python tools/prof_summary.py <db> $images   (MI355X, final round-3 kernels)"; python $R/tools/prof_summary.py $(db $O/tmp_$name) $images; } > $O/${TAG}_kernel_trace_$name.txt 2>&1
    rm -rf $O/tmp_$name
}
trace bench_b32_570m 96 python $R/bench.py --batch 32 --steps 1 --warmup 1 $COMMON
head -8 $O/${TAG}_kernel_trace_bench_b32_570m.txt | cut -c1-140
trace config3_b64_64x64 192 python $R/bench.py --batch 64 --grid 64 --sample-steps 12 --steps 1 --warmup 1 $COMMON
head -8 $O/${TAG}_kernel_trace_config3_b64_64x64.txt | cut -c1-140
