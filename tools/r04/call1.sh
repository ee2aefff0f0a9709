#!/bin/bash
# round 4, GPU call 1: the 256x128 throughput-regime tile (id 36) -- parity tests, isolated sweep against the round-3 tiles, in-model A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "36 or big_tile or stream_k or heuristic" > $O/pytest_big.txt 2>&1
tail -5 $O/pytest_big.txt
for apro in 0 1 2; do
  timeout 400 python tools/gemm_tune.py --cfgs 10,18,36 --apro $apro --only "c3 ,b32 " > $O/tune_apro$apro.txt 2>&1
done
timeout 300 python tools/gemm_tune.py --cfgs 10,18,36 --act 1 --only "c3 L1 mlp1,c3 L0 mlp1,b32 L1 mlp1,b32 L0 mlp1" > $O/tune_gelu.txt 2>&1
grep -h "best" $O/tune_*.txt | cut -c1-250
for mode in 0 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra --batch 32 --steps 3 --warmup 1 --hook gemm_big=$mode > $O/bench_b32_big$mode.json 2> $O/bench_b32_big$mode.err
  timeout 400 python bench.py --no-cpu-baseline --no-extra --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1 --hook gemm_big=$mode > $O/bench_c3_big$mode.json 2> $O/bench_c3_big$mode.err
done
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04c1"
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print(os.path.basename(f), "img/s %.2f ms/img %.2f exec TF %.1f gemm_ms %.1f" % (j["value"], j["single_image_ms"], r["executed_tflops"], r["gemm_ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
