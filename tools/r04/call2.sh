#!/bin/bash
# round 4, GPU call 2: sustained-rate probe (why are in-model GEMMs ~10 % below their isolated sweeps?) + the LayerNorm-fold numerics tests + the whole GPU suite on the new rowstat format
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c2
mkdir -p $O
cd $R
timeout 120 python tools/sustained_probe.py --cfg 18 --seconds 8 > $O/sustained_cfg18.txt 2>&1
timeout 120 python tools/sustained_probe.py --cfg 36 --seconds 5 > $O/sustained_cfg36.txt 2>&1
timeout 120 python tools/sustained_probe.py --vendor --seconds 8 > $O/sustained_vendor.txt 2>&1
timeout 120 python tools/sustained_probe.py --cfg 18 --seconds 5 --act 1 > $O/sustained_cfg18_gelu.txt 2>&1
tail -12 $O/sustained_cfg18.txt; tail -3 $O/sustained_cfg36.txt; tail -12 $O/sustained_vendor.txt; tail -3 $O/sustained_cfg18_gelu.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -s -p no:cacheprovider -k "layernorm_fold or big_tile" > $O/pytest_ln.txt 2>&1
grep -v amdgpu $O/pytest_ln.txt | tail -40
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -8 $O/pytest_all.txt
