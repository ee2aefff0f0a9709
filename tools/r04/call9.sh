#!/bin/bash
# round 4, GPU call 9: per-kernel same-box comparison of the round-3 tree and HEAD at batch 1 (kernel traces), + fused-tail parity tests after the cheaper scoring
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c9
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sample.py tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "tail or fused or margin or philox or gumbel or shard" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for t in r03 head; do
  D=$R; [ $t = r03 ] && D=$R/ab_r03
  ( cd $D && rocprofv3 --kernel-trace --output-format rocpd -d $O/tr_${t}_$rep -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-graph > $O/trace_${t}_$rep.log 2>&1 )
  python $R/tools/prof_summary.py $(find $O/tr_${t}_$rep -name "*.db" | head -1) 6 > $O/trace_${t}_$rep.txt 2>&1
  rm -rf $O/tr_${t}_$rep
done
done
python - <<'PY'
import os, re
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04c9"
def load(f):
    d = {}
    for line in open(f):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)%\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m: d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(5)))
    return d
for rep in (1, 2):
    a, b = load(O + "/trace_r03_%d.txt" % rep), load(O + "/trace_head_%d.txt" % rep)
    print("rep", rep, "total us r03 %.0f head %.0f" % (sum(v[1] for k, v in a.items() if "rocclr" not in k and "permute" not in k), sum(v[1] for k, v in b.items() if "rocclr" not in k and "permute" not in k)))
    for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[1] + b.get(k, (0, 0, 0))[1]))[:22]:
        x, y = a.get(k, (0, 0.0, 0.0)), b.get(k, (0, 0.0, 0.0))
        print("%-72s r03 %5d x %7.2f = %9.0f | head %5d x %7.2f = %9.0f | d %+7.0f" % (k[:72], x[0], x[2], x[1], y[0], y[2], y[1], y[1] - x[1]))
PY
