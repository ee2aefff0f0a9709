#!/bin/bash
# Round 3 final GPU call: the whole GPU suite, the round's rocprofv3 evidence (tools/collect_profiles.sh), then the bench line with in-date traffic.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1; tail -2 $O/build_smoke.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest_full.log 2>&1
tail -4 $O/pytest_full.log
bash tools/collect_profiles.sh r03 > $O/collect.log 2>&1
cp $R/gpurun_out/r03_profiles/r03_pmc_traffic_*.json $R/profiles/ 2>/dev/null
cd $R
timeout 900 python bench.py > $O/r03_bench_line.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open("$O/r03_bench_line.json").read().strip().splitlines()[-1])
print("bench: %.3f ms/image %.2f img/s; frac %.3f exec %.3f traffic %s; cpu %s" % (j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["executed_frac"], j["roofline"]["traffic"], j["cpu_baseline"]["value"]))
for t in j["throughput"]: print(" ", t.get("batch"), t.get("grid"), t.get("images_per_sec"), t.get("roofline",{}).get("executed_frac"), t.get("roofline",{}).get("traffic"))
PY
ls $R/gpurun_out/r03_profiles | head -40
