#!/bin/bash
# Round 3 code freeze: PMC traffic files re-collected on the final kernel sources, then the bench line that cites them.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_final2
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
python -m pytest tests/test_gpu_sample.py tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "fused_head or benchmarked or prologues or direct_to_lds" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra --no-graph"
traffic() {  # name, batch grid sample_steps, command...
    local name=$1 b=$2 g=$3 s=$4; shift 4
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $O/tmp_f_$name -- "$@" > $O/log_pmc_fetch_$name.txt 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $O/tmp_w_$name -- "$@" > $O/log_pmc_write_$name.txt 2>&1
    python $R/tools/pmc_traffic.py $O/tmp_f_$name $O/tmp_w_$name $O/r03_pmc_traffic_$name.json $b $g $s > $O/log_pmc_traffic_$name.txt 2>&1
    rm -rf $O/tmp_f_$name $O/tmp_w_$name
}
traffic b1 1 32 8 python $R/bench.py --steps 2 --warmup 1 $COMMON
traffic b32 32 32 8 python $R/bench.py --batch 32 --steps 1 --warmup 1 $COMMON
traffic config3 64 64 12 python $R/bench.py --batch 64 --grid 64 --sample-steps 12 --steps 1 --warmup 0 $COMMON
cp $O/r03_pmc_traffic_*.json $R/profiles/
cd $R
timeout 900 python bench.py > $O/r03_bench_line.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open("$O/r03_bench_line.json").read().strip().splitlines()[-1])
print("bench: %.3f ms/image %.2f img/s; frac %.3f exec %.3f traffic %s; cpu %s" % (j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["executed_frac"], j["roofline"]["traffic"], j["cpu_baseline"]["value"]))
for t in j["throughput"]: print(" ", t.get("batch"), t.get("grid"), t.get("images_per_sec"), t.get("roofline",{}).get("executed_frac"), t.get("roofline",{}).get("traffic"))
PY
