#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "prologues or direct_to_lds or heuristic" 2>&1 | tail -2
python -m pytest tests/test_gpu_unet.py tests/test_gpu_fastmode.py -q -x -p no:cacheprovider 2>&1 | tail -2
for lp in 0 1; do
  for w in "b1 --steps 10" "b32 --batch 32 --steps 3" "c3 --batch 64 --grid 64 --sample-steps 12 --steps 1"; do set -- $w; n=$1; shift
    PAELLA_LN_AS_PLAIN=$lp python bench.py --no-cpu-baseline --no-extra --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LN folded, plain rules $lp $n: %.3f ms/step, %.2f img/s, exec frac %.3f' % (j['ms_per_step'], j['value'], j['roofline']['executed_frac']))"
  done
done
