#!/bin/bash
# Round 3, GPU call 2: ring vs legacy per prologue with full per-variant numbers; kernel trace of the bench with the ring tile in the heuristic.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c2
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
SK="L0 mlp1,L0 mlp2,L1 mlp1,L1 mlp2,L1 qkv,L1 out,L2 mlp1,L2 mlp2,L2 qkv,L2 out,pfx L0,pfx L1,up1,down1,clf,up2,down2,embed"
CF="5,19,24,4,30,31,32"
timeout 300 python tools/gemm_tune.py --only "$SK" --cfgs $CF --out $O/tune_apro0.json > $O/tune_apro0.txt 2>&1
timeout 200 python tools/gemm_tune.py --apro 1 --only "L0 mlp2,L1 mlp2,L2 mlp2,pfx L0 mlp2,pfx L1 mlp2" --cfgs $CF --out $O/tune_apro1.json > $O/tune_apro1.txt 2>&1
timeout 200 python tools/gemm_tune.py --apro 2 --only "L1 qkv,L2 qkv,up1,up2,clf" --cfgs $CF --out $O/tune_apro2.json > $O/tune_apro2.txt 2>&1
python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "prologues or every_tile" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for ring in 30 0; do
  PAELLA_GEMM_RING=$ring rocprofv3 --kernel-trace --output-format rocpd -d $O/tr_$ring -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-graph > $O/trace_$ring.log 2>&1
  python $R/tools/prof_summary.py $(find $O/tr_$ring -name "*.db" | head -1) 4 > $O/kernel_trace_ring$ring.txt 2>&1
  rm -rf $O/tr_$ring
  head -14 $O/kernel_trace_ring$ring.txt | cut -c1-200
done
cd $R
grep -h -A1 "best" $O/tune_apro0.txt | cut -c1-250
echo ---- apro1; grep -h -A1 "best" $O/tune_apro1.txt | cut -c1-250
echo ---- apro2; grep -h -A1 "best" $O/tune_apro2.txt | cut -c1-250
