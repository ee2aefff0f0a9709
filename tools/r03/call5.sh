#!/bin/bash
# Round 3, GPU call 5: full GPU suite + parity report (-s) + bench with extras; A/B of the fused-tail tile.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c5
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/pytest_full.log 2>&1
tail -8 $O/pytest_full.log
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "parity or vs_oracle or vs_reference or closed_loop or benchmarked or geometry or train_step or prompts_to_image or graph_sampler or reproduces" > $O/parity_report_raw.txt 2>&1
grep -v "^\s*$" $O/parity_report_raw.txt | grep -iv "amdgpu.ids" | tail -60
for tt in 9 14; do
  PAELLA_GEMM_TAIL_TILE=$tt timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > $O/bench_b1_tail$tt.json 2>/dev/null
  PAELLA_GEMM_TAIL_TILE=$tt timeout 300 python bench.py --no-cpu-baseline --no-extra --batch 32 --steps 3 --warmup 1 > $O/bench_b32_tail$tt.json 2>/dev/null
  PAELLA_GEMM_TAIL_TILE=$tt timeout 400 python bench.py --no-cpu-baseline --no-extra --batch 64 --grid 64 --sample-steps 12 --steps 2 --warmup 1 > $O/bench_c3_tail$tt.json 2>/dev/null
  python - <<PY
import json
for n in ("b1","b32","c3"):
    try:
        j=json.loads(open("$O/bench_%s_tail$tt.json"%n).read().strip().splitlines()[-1])
        print("tail tile $tt %s: %.3f ms/step, %.2f img/s, gemm exec frac %.3f" % (n, j["ms_per_step"], j["value"], j["roofline"]["executed_frac"]))
    except Exception as e:
        print("tail tile $tt", n, "failed", e)
PY
done
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err
tail -c 3000 $O/bench_line.json
