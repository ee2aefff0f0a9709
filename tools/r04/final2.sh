#!/bin/bash
# round 4, last check of the committed tree: whole GPU suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_final2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -3 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench_line.json
python - <<'PY'
import json, os
j = json.load(open(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_final2/bench_line.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["executed_frac"], j["roofline"]["traffic"], [(t["batch"], t["images_per_sec"], t["roofline"]["traffic"]) for t in j["throughput"]], j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
PY
