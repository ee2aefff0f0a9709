#!/bin/bash
# round 4, GPU call 6: why is attention_kernel<5,true> 7.5 us per launch at batch 1 (5.6 in round 2, same kernel source)?  Kernel traces of the batch-1 bench with the round-2 GEMM kernels
# (hook gemm_ring=0) and with the default ring tiles; + shader clock / power while BASELINE configs[2] runs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ring in 30 0; do
  rocprofv3 --kernel-trace --output-format rocpd -d $O/tr_$ring -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-graph --hook gemm_ring=$ring > $O/trace_$ring.log 2>&1
  python $R/tools/prof_summary.py $(find $O/tr_$ring -name "*.db" | head -1) 5 > $O/trace_ring$ring.txt 2>&1
  rm -rf $O/tr_$ring
  echo "== ring $ring"; head -12 $O/trace_ring$ring.txt | cut -c1-160
done
cd $R
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -1; sleep 0.5; done ) > $O/smi_c3.txt &
timeout 300 python bench.py --no-cpu-baseline --no-extra --batch 64 --grid 64 --sample-steps 12 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
wait
cut -d, -f6,10 $O/smi_c3.txt | tr '\n' ' '
python -c "
import json; j=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1]); print('c3', j['value'], j['roofline']['executed_tflops'])"
