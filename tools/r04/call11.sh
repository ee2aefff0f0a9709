#!/bin/bash
# round 4, GPU call 11: fused head + tail on the 64x64 direct-to-LDS tile (four workgroups per CU) against the 128x64 and 128x128 tiles; whole suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c11
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt
run() {  # tag, args...
  local tag=$1; shift
  timeout 400 python bench.py --no-cpu-baseline --no-extra "$@" 2> $O/$tag.err | tail -1 > $O/$tag.json
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-22s img/s %7.2f  ms/img %7.3f  exec TF %6.1f  gemm_ms %8.1f  hooks %s" % (sys.argv[2], j["value"], j["single_image_ms"], r["executed_tflops"], r["gemm_ms_per_step"], j.get("test_hooks")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for t in 18 14 9; do
  run b1_tail$t --steps 20 --warmup 3 --hook gemm_tail_tile=$t
  run b32_tail$t --batch 32 --steps 3 --warmup 1 --hook gemm_tail_tile=$t
  run c3_tail$t --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1 --hook gemm_tail_tile=$t
done
