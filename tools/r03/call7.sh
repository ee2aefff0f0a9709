#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c7
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 12 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label: %.3f ms/image, gemm ms %.2f' % (j['ms_per_step'], j['roofline']['gemm_ms_per_step']))"
}
run "rule 9" PAELLA_GEMM_RULE=9
run "rule 9+16 (U/8)" PAELLA_GEMM_RULE=25
run "rule 9+32 (U/6)" PAELLA_GEMM_RULE=41
run "rule 9+64 (1536 resident)" PAELLA_GEMM_RULE=73
run "rule 9+64+16" PAELLA_GEMM_RULE=89
run "ring 31 rule 1" PAELLA_GEMM_RING=31 PAELLA_GEMM_RULE=1
run "rule 9 again" PAELLA_GEMM_RULE=9
