#!/bin/bash
# round 4, GPU call 12: branch-free / division-free GEMM prologue (one kernel-argument fetch, first operand fetch after ~100 straight-line instructions): same-box A/B against the round-3 tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "gemm" > $O/pytest_gemm.txt 2>&1
tail -3 $O/pytest_gemm.txt
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
cd /tmp && export TMPDIR=/tmp
for t in r03 head; do
  D=$R; [ $t = r03 ] && D=$R/ab_r03
  ( cd $D && rocprofv3 --kernel-trace --output-format rocpd -d $O/tr_${t} -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-graph > $O/trace_${t}.log 2>&1 )
  python $R/tools/prof_summary.py $(find $O/tr_${t} -name "*.db" | head -1) 6 > $O/trace_${t}.txt 2>&1
  rm -rf $O/tr_${t}
done
python - <<'PY'
import os, re
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04c12"
def load(f):
    d = {}
    for line in open(f):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)%\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m: d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(5)))
    return d
a, b = load(O + "/trace_r03.txt"), load(O + "/trace_head.txt")
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[1] + b.get(k, (0, 0, 0))[1]))[:12]:
    x, y = a.get(k, (0, 0.0, 0.0)), b.get(k, (0, 0.0, 0.0))
    print("%-72s r03 %5d x %7.2f | head %5d x %7.2f | d/launch %+6.2f us" % (k[:72], x[0], x[2], y[0], y[2], y[2] - x[2]))
PY
