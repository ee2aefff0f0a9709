#!/bin/bash
# round 4, GPU call 3: LayerNorm-fold numerics (centred partials + guard), whole GPU suite, large-shape sweep incl. the 64x64 ring tile (34) and the 128x128 4-wave direct-to-LDS twin (0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -s -p no:cacheprovider -k "layernorm_fold or big_tile" 2>&1 | grep -v amdgpu > $O/pytest_ln.txt
grep "cfg\|passed\|failed" $O/pytest_ln.txt | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -15 $O/pytest_all.txt
for apro in 0 1; do
  timeout 400 python tools/gemm_tune.py --cfgs 0,10,18,34,36 --apro $apro --only "c3 L1,c3 L0,b32 L1 mlp,b32 L0" > $O/tune_apro$apro.txt 2>&1
done
grep -h "best" $O/tune_*.txt | cut -c1-260
for gm in 0 4 16; do
  echo "raster $gm"; timeout 200 python tools/gemm_tune.py --cfgs 18,36 --raster $gm --only "c3 L1 mlp" 2>&1 | grep best | cut -c1-200
done
