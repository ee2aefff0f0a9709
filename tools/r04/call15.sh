#!/bin/bash
# round 4, GPU call 15: epilogue operand loads issued as one batch (bias / residual / scale-shift): GEMM parity tests, same-box A/B against the round-3 tree, per-kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c15
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -x -p no:cacheprovider > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
run() {  # tag, dir, args...
  local tag=$1 dir=$2; shift 2
  ( cd $dir && timeout 400 python bench.py --no-cpu-baseline --no-extra "$@" 2> $O/$tag.err | tail -1 > $O/$tag.json )
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-22s img/s %7.2f  ms/img %7.3f  exec TF %6.1f  gemm_ms %8.1f" % (sys.argv[2], j["value"], j["single_image_ms"], r["executed_tflops"], r["gemm_ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do
  run r03_b1_$rep $R/ab_r03 --steps 20 --warmup 3
  run head_b1_$rep $R --steps 20 --warmup 3
done
run r03_b32 $R/ab_r03 --batch 32 --steps 3 --warmup 1
run head_b32 $R --batch 32 --steps 3 --warmup 1
