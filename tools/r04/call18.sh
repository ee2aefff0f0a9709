#!/bin/bash
# round 4, GPU call 18: the PARKED opt-in bf16-operand mode re-measured on the round-4 tree (numbers in DESIGN section 8 were round 1's): bench line + the fast-mode tests with their printed deviations
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c18
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fastmode.py -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu | grep -i "flip\|dev\|bf16\|passed\|failed\|argmax" | cut -c1-300 | tee $O/fastmode_tests.txt
timeout 600 python bench.py --gemm bf16 --no-cpu-baseline 2> $O/bench_bf16.err | tail -1 > $O/bench_line_bf16.json
python - <<'PY'
import json, os
j = json.load(open(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04c18/bench_line_bf16.json"))
print("bf16 mode: batch 1 %.2f ms/image (%.1f images/s), executed %.1f TFLOP/s; " % (j["single_image_ms"], j["value"], j["roofline"]["executed_tflops"]) +
      "; ".join("batch %d grid %d: %.1f images/s, executed %.1f TFLOP/s" % (t["batch"], t["grid"], t["images_per_sec"], t["roofline"]["executed_tflops"]) for t in j["throughput"]))
PY
