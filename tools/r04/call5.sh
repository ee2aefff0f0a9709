#!/bin/bash
# round 4, GPU call 5: wave-staggered 256x128 tile (stagger 0 / 1 / 2) against the 64x64 tile; RCCL check; parity of tile 36
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py -q -x -s -p no:cacheprovider -k "36 or big_tile or multi_m or rccl" 2>&1 | grep -v amdgpu | tail -5 | cut -c1-400
for sg in 0 1 2; do
  echo "== stagger $sg"
  timeout 300 python - <<PY 2>&1 | grep best | cut -c1-220
import sys, subprocess
sys.argv = ["gemm_tune.py", "--cfgs", "18,36", "--only", "c3 L1 mlp,c3 L0 mlp1,c3 L1 qkv,b32 L1 mlp1"]
sys.path.insert(0, "tools")
from paella_amd import _lib
_lib.load().paella_test_gemm_big_stagger($sg)
import runpy
runpy.run_path("tools/gemm_tune.py", run_name="__main__")
PY
done
echo "== GRN prologue, stagger 1"
timeout 300 python tools/gemm_tune.py --cfgs 10,18,36 --apro 1 --only "c3 L1 mlp2,c3 L0 mlp2" 2>&1 | grep best | cut -c1-220
echo "== gelu epilogue, stagger 1"
timeout 300 python tools/gemm_tune.py --cfgs 18,36 --act 1 --only "c3 L1 mlp1,c3 L0 mlp1" 2>&1 | grep best | cut -c1-220
