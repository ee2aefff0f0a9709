#!/bin/bash
# refresh the batch-1 trace / counters / per-shape table and the parity report on the final kernels (after the LayerNorm fold)
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_final3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra --no-graph"
db() { find $1 -name "*.db" | head -1; }
rocprofv3 --kernel-trace --output-format rocpd -d $O/tmp_t -- python $R/bench.py --steps 4 --warmup 1 $COMMON > $O/log_trace.txt 2>&1
{ echo "# rocprofv3 --kernel-trace --output-format rocpd -- python bench.py --steps 4 --warmup 1 $COMMON ; python tools/prof_summary.py <db> 6   (MI355X, final round-3 kernels)"; python $R/tools/prof_summary.py $(db $O/tmp_t) 6; } > $O/${TAG}_kernel_trace_bench_b1_570m.txt 2>&1
rm -rf $O/tmp_t
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format rocpd -d $O/tmp_m -- python $R/bench.py --steps 2 --warmup 1 $COMMON > $O/log_mfma.txt 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python bench.py --steps 2 --warmup 1 $COMMON ; python tools/pmc_summary.py <db> gemm_nt_kernel   (final round-3 kernels)"
  echo "# per-launch averages; matrix-core utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)"
  python $R/tools/pmc_summary.py $(db $O/tmp_m) gemm_nt_kernel; } > $O/${TAG}_pmc_mfma_busy_b1.txt 2>&1
rm -rf $O/tmp_m
cd $R
python tools/gemm_by_shape.py 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_b1.txt
{ echo "# python -m pytest tests -m gpu -q -s -k 'parity or vs_oracle or vs_reference or closed_loop or benchmarked or geometry or train_step or prompts_to_image or graph_sampler or reproduces or grn_finished or lds_kernel'   (MI355X, final round-3 kernels)"
  python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "parity or vs_oracle or vs_reference or closed_loop or benchmarked or geometry or train_step or prompts_to_image or graph_sampler or reproduces or grn_finished or lds_kernel" 2>&1 | grep -v "amdgpu.ids" | grep -v "^\s*$"; } > $O/${TAG}_parity_report.txt
head -8 $O/${TAG}_kernel_trace_bench_b1_570m.txt | cut -c1-150; tail -3 $O/${TAG}_parity_report.txt
