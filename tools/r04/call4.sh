#!/bin/bash
# round 4, GPU call 4: RCCL path at world size 1, LayerNorm-fold tests at threshold 4, cold-vs-warm operand probe, whole suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu > $O/pytest_dist.txt
tail -12 $O/pytest_dist.txt | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_ops.py -q -s -p no:cacheprovider -k "layernorm_fold or big_tile or multi_m" 2>&1 | grep -v amdgpu > $O/pytest_ln.txt
grep "fold always\|operand-side\|guarded\|passed\|failed" $O/pytest_ln.txt | cut -c1-300
timeout 600 python tools/gemm_cold_probe.py 2>&1 | grep -v amdgpu > $O/cold_probe.txt
cat $O/cold_probe.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -8 $O/pytest_all.txt
