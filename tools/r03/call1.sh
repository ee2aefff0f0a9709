#!/bin/bash
# Round 3, GPU call 1: the whole GPU suite (new parity tests included), then the LDS-DMA ring tiles against the shipped batch-1 tiles.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c1
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest_full.log 2>&1
tail -15 $O/pytest_full.log
SK="L0 mlp1,L0 mlp2,L1 mlp1,L1 mlp2,L1 qkv,L1 out,L2 mlp1,L2 mlp2,L2 qkv,L2 out,pfx L0,pfx L1,up1,down1,clf"
CF="5,19,24,4,30,31,32,33,34,35"
timeout 300 python tools/gemm_tune.py --only "$SK" --cfgs $CF > $O/tune_apro0.txt 2>&1
timeout 200 python tools/gemm_tune.py --apro 1 --only "mlp2" --cfgs $CF > $O/tune_apro1.txt 2>&1
timeout 200 python tools/gemm_tune.py --apro 2 --only "qkv,up1,clf,L1 out" --cfgs $CF > $O/tune_apro2.txt 2>&1
for ring in 0 30 31 32; do
  PAELLA_GEMM_RING=$ring timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > $O/bench_ring$ring.json 2> $O/bench_ring$ring.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/bench_ring$ring.json").read().strip().splitlines()[-1])
    print("ring $ring: %.3f ms/image, %.2f img/s, gemm ms %.2f, launches %d, exec frac %.3f" % (j["ms_per_step"], j["value"], j["roofline"]["gemm_ms_per_step"], j["roofline"]["launches_per_step"], j["roofline"]["executed_frac"]))
except Exception as e:
    print("ring $ring: failed", e)
PY
done
grep -h "best" $O/tune_apro0.txt | cut -c1-230
echo ---- apro1; grep -h "best" $O/tune_apro1.txt | cut -c1-230
echo ---- apro2; grep -h "best" $O/tune_apro2.txt | cut -c1-230
