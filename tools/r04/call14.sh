#!/bin/bash
# round 4, GPU call 14: GlobalResponseNorm finished inside the MLP GEMMs at the 16-row-sample level (APRO 4) against the finalize launch, in the model (bench.py --hook grn_fuse=0/1); autograd test re-run
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c14
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_unet.py -q -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2 3; do
for f in 1 0; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 --hook grn_fuse=$f 2> $O/b1_grn${f}_$rep.err | tail -1 > $O/b1_grn${f}_$rep.json
  python - $O/b1_grn${f}_$rep.json "grn_fuse=$f rep $rep" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print("%-18s ms/image %7.3f  GEMM launches %4d  GEMM ms/step %6.2f" % (sys.argv[2], j["single_image_ms"], r["launches_per_step"], r["gemm_ms_per_step"]))
PY
done
done
