#!/bin/bash
# round 4, GPU call 17: attention kernels with two S accumulators and the PV MFMAs ordered for independence (no dependent back-to-back MFMAs): parity tests, the configs[4]-shape probe, batch-1 trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c17
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -x -p no:cacheprovider -k "attention or attn or forward or geometry" 2>&1 | tail -2
timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu | tee $O/attn_probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d $O/tr -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-graph > $O/trace.log 2>&1
python $R/tools/prof_summary.py $(find $O/tr -name "*.db" | head -1) 6 > $O/trace_b1.txt 2>&1
rm -rf $O/tr
grep "attention\|dwconv_ln_block_kernel<2, false>\|GPU kernel time" $O/trace_b1.txt | cut -c1-160
