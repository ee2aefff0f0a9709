#!/bin/bash
# Re-collects the three HBM traffic files bench.py reads (they are stamped with the GEMM kernel sources' hash and refused when stale) and the default bench line.
# Run on a GPU box from the repo root: bash tools/recollect_traffic.sh r05
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra --no-graph"
traffic() {  # name, batch grid sample_steps, command...
    local name=$1 b=$2 g=$3 s=$4; shift 4
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $O/tmp_f_$name -- "$@" > $O/log_pmc_fetch_$name.txt 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $O/tmp_w_$name -- "$@" > $O/log_pmc_write_$name.txt 2>&1
    python $R/tools/pmc_traffic.py $O/tmp_f_$name $O/tmp_w_$name $O/${TAG}_pmc_traffic_$name.json $b $g $s > $O/log_pmc_traffic_$name.txt 2>&1
    rm -rf $O/tmp_f_$name $O/tmp_w_$name
}
traffic b1 1 32 8 python $R/bench.py --steps 2 --warmup 1 $COMMON
traffic b32 32 32 8 python $R/bench.py --batch 32 --steps 1 --warmup 1 $COMMON
traffic config3 64 64 12 python $R/bench.py --batch 64 --grid 64 --sample-steps 12 --steps 1 --warmup 0 $COMMON
cp $O/${TAG}_pmc_traffic_*.json $R/profiles/ 2>/dev/null
cd $R
python bench.py 2> $O/log_bench_line.txt | tail -1 > $O/${TAG}_bench_line.json
ls -la $O | tail -8
