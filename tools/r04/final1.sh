#!/bin/bash
# round 4, final collection: whole GPU suite, then tools/collect_profiles.sh r04 (traces, PMC traffic / matrix-core busy, per-shape tables, parity report, LayerNorm-fold curve, RCCL check, bench line)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_profiles
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -4 $O/pytest_all.txt
timeout 2400 bash tools/collect_profiles.sh r04 > $O/collect.log 2>&1
tail -5 $O/collect.log
cat $O/r04_bench_line.json | cut -c1-600
