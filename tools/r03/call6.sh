#!/bin/bash
# in-model A/B of the ring workgroup-count rules (microbenchmarks with warm activations did not predict the model)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c6
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for rule in 0 1 9 3 11 15 7; do
  PAELLA_GEMM_RULE=$rule timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 12 --warmup 2 > $O/bench_rule$rule.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("$O/bench_rule$rule.json").read().strip().splitlines()[-1])
print("rule mask $rule: %.3f ms/image, gemm ms %.2f" % (j["ms_per_step"], j["roofline"]["gemm_ms_per_step"]))
PY
done
PAELLA_GEMM_RING=0 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 12 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('legacy (ring 0): %.3f ms/image, gemm ms %.2f' % (j['ms_per_step'], j['roofline']['gemm_ms_per_step']))"
python -m pytest tests/test_gpu_sample.py -q -s -p no:cacheprovider -k "benchmarked or fused_head or start_tokens" 2>&1 | grep -v amdgpu | tail -8
