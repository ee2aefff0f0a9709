#!/bin/bash
# round 4, GPU call 10: does the size of the kernel-argument segment cost launch latency?  HEAD vs HEAD + 64 bytes of padding in GemmArgs (ab_pad/), batch-1 kernel traces on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for t in head pad; do
  D=$R; [ $t = pad ] && D=$R/ab_pad
  ( cd $D && rocprofv3 --kernel-trace --output-format rocpd -d $O/tr_${t}_$rep -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-graph > $O/trace_${t}_$rep.log 2>&1 )
  python $R/tools/prof_summary.py $(find $O/tr_${t}_$rep -name "*.db" | head -1) 6 > $O/trace_${t}_$rep.txt 2>&1
  rm -rf $O/tr_${t}_$rep
done
done
python - <<'PY'
import os, re
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04c10"
def load(f):
    d = {}
    for line in open(f):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)%\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m: d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(5)))
    return d
for rep in (1, 2):
    a, b = load(O + "/trace_head_%d.txt" % rep), load(O + "/trace_pad_%d.txt" % rep)
    for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[1] + b.get(k, (0, 0, 0))[1]))[:8]:
        x, y = a.get(k, (0, 0.0, 0.0)), b.get(k, (0, 0.0, 0.0))
        print("%-72s head %5d x %7.2f | pad %5d x %7.2f | d/launch %+6.2f us" % (k[:72], x[0], x[2], y[0], y[2], y[2] - x[2]))
PY
