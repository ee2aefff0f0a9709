#!/bin/bash
# round 4, GPU call 7: block-major conditioning cache (attention launch time at batch 1), LayerNorm row statistics finished by a pre-pass in the throughput regime; whole suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c7
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d $O/tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-graph > $O/trace.log 2>&1
python $R/tools/prof_summary.py $(find $O/tr -name "*.db" | head -1) 5 > $O/trace_b1.txt 2>&1
rm -rf $O/tr
head -14 $O/trace_b1.txt | cut -c1-160
cd $R
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --no-cpu-baseline --no-extra --batch 32 --steps 3 --warmup 1 > $O/bench_b32.json 2> $O/bench_b32.err
timeout 400 python bench.py --no-cpu-baseline --no-extra --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04c7"
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print(os.path.basename(f), "img/s %.2f ms/img %.3f exec TF %.1f gemm_ms %.1f launches %d" % (j["value"], j["single_image_ms"], r["executed_tflops"], r["gemm_ms_per_step"], r["launches_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
python tools/gemm_cold_probe.py 2>&1 | grep -v amdgpu | grep "LN\|case"
