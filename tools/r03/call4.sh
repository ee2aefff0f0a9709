#!/bin/bash
# does placing kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) shorten the per-launch fixed cost?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c4
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
SK="L0 mlp1,L1 mlp1,L1 out,L2 out"
for kv in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$kv"
  HIP_FORCE_DEV_KERNARG=$kv timeout 200 python tools/gemm_tune.py --only "$SK" --cfgs 19,30 2>&1 | grep "best" | grep -v "^c3\|^b32\|^b8\|pfx" | cut -c1-200
  for ring in 0 30; do
    HIP_FORCE_DEV_KERNARG=$kv PAELLA_GEMM_RING=$ring timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > $O/bench_kv${kv}_ring$ring.json 2> $O/bench_kv${kv}_ring$ring.err
    python - <<PY
import json
j=json.loads(open("$O/bench_kv${kv}_ring$ring.json").read().strip().splitlines()[-1])
print("kernarg-dev $kv ring $ring (graph): %.3f ms/image, gemm ms %.2f" % (j["ms_per_step"], j["roofline"]["gemm_ms_per_step"]))
PY
  done
  HIP_FORCE_DEV_KERNARG=$kv PAELLA_GEMM_RING=30 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 --no-graph > $O/bench_kv${kv}_eager.json 2>/dev/null
  python -c "
import json
j=json.loads(open('$O/bench_kv${kv}_eager.json').read().strip().splitlines()[-1]); print('kernarg-dev $kv ring 30 (eager): %.3f ms/image' % j['ms_per_step'])"
done
