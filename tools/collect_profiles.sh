#!/bin/bash
# Collects the round's rocprofv3 evidence on a GPU box (run from the repo root: bash tools/collect_profiles.sh r02).
# Output: gpurun_out/<tag>_profiles/ -- kernel-trace summaries (batch 1, batch 32, BASELINE configs[2]), the HBM traffic files
# bench.py reads (tools/pmc_traffic.py; stamped with the kernel-source hash), matrix-core busy counters.  Counters are collected
# in passes of their own with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra --no-graph"
W_B1="python $R/bench.py"
W_B32="python $R/bench.py --batch 32"
W_B128="python $R/bench.py --batch 128"
W_C3="python $R/bench.py --batch 64 --grid 64 --sample-steps 12"
db() { find $1 -name "*.db" | head -1; }

trace() {  # name, images, command...
    local name=$1 images=$2; shift 2
    rocprofv3 --kernel-trace --output-format rocpd -d $O/tmp_$name -- "$@" > $O/log_trace_$name.txt 2>&1
    { echo "# rocprofv3 --kernel-trace --output-format rocpd -- $* ; python tools/prof_summary.py <db> $images   (MI355X)"; python $R/tools/prof_summary.py $(db $O/tmp_$name) $images; } > $O/${TAG}_kernel_trace_$name.txt 2>&1
    rm -rf $O/tmp_$name
}
trace bench_b1_570m 6 $W_B1 --steps 4 --warmup 1 $COMMON
trace bench_b32_570m 96 $W_B32 --steps 1 --warmup 1 $COMMON
trace bench_b128_570m 384 $W_B128 --steps 1 --warmup 1 $COMMON
trace config3_b64_64x64 192 $W_C3 --steps 1 --warmup 1 $COMMON
# BASELINE configs[3] / configs[4] per-GPU shares as bench.py times them (released-size 1B model, ByT5 256 + CLIP text + CLIP image; configs[4] = the inpainting path); eager, 2 batches each
W_C4="python $R/bench.py --model 1b --batch 32 --grid 64 --sample-steps 12 --s-byt5 256 --clip-image 1"
W_C5="python $R/bench.py --model 1b --batch 16 --grid 128 --sample-steps 12 --s-byt5 256 --clip-image 1 --inpaint"
trace configs3_share_1b_b32_64x64 64 $W_C4 --steps 1 --warmup 0 $COMMON
trace configs4_share_1b_b16_128x128_inpaint 32 $W_C5 --steps 1 --warmup 0 $COMMON
# the opt-in bf16 fast mode (outside the parity contract): the same three 570M workloads
trace bf16_fastmode_b1_570m 6 $W_B1 --gemm bf16 --steps 4 --warmup 1 $COMMON
trace bf16_fastmode_b32_570m 96 $W_B32 --gemm bf16 --steps 1 --warmup 1 $COMMON
trace bf16_fastmode_config3_b64_64x64 192 $W_C3 --gemm bf16 --steps 1 --warmup 1 $COMMON

traffic() {  # name, batch grid sample_steps, command...
    local name=$1 b=$2 g=$3 s=$4; shift 4
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $O/tmp_f_$name -- "$@" > $O/log_pmc_fetch_$name.txt 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $O/tmp_w_$name -- "$@" > $O/log_pmc_write_$name.txt 2>&1
    python $R/tools/pmc_traffic.py $O/tmp_f_$name $O/tmp_w_$name $O/${TAG}_pmc_traffic_$name.json $b $g $s > $O/log_pmc_traffic_$name.txt 2>&1
    rm -rf $O/tmp_f_$name $O/tmp_w_$name
}
traffic b1 1 32 8 $W_B1 --steps 2 --warmup 1 $COMMON
traffic b32 32 32 8 $W_B32 --steps 1 --warmup 1 $COMMON
traffic b64 64 32 8 python $R/bench.py --batch 64 --steps 1 --warmup 0 $COMMON
traffic b128 128 32 8 $W_B128 --steps 1 --warmup 0 $COMMON
traffic config3 64 64 12 $W_C3 --steps 1 --warmup 0 $COMMON
# the 1B shares (no `traffic: null` in the line); pmc_traffic.py gets the model name and the remaining flags as its 7th / 8th argument
traffic1b() {  # name, batch grid sample_steps, more flags (quoted), command...
    local name=$1 b=$2 g=$3 s=$4 more=$5; shift 5
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $O/tmp_f_$name -- "$@" > $O/log_pmc_fetch_$name.txt 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $O/tmp_w_$name -- "$@" > $O/log_pmc_write_$name.txt 2>&1
    python $R/tools/pmc_traffic.py $O/tmp_f_$name $O/tmp_w_$name $O/${TAG}_pmc_traffic_$name.json $b $g $s 1b "$more" > $O/log_pmc_traffic_$name.txt 2>&1
    rm -rf $O/tmp_f_$name $O/tmp_w_$name
}
traffic1b configs3_share_1b 32 64 12 "--s-byt5 256 --clip-image 1" $W_C4 --steps 1 --warmup 0 $COMMON
traffic1b configs4_share_1b 16 128 12 "--s-byt5 256 --clip-image 1 --inpaint" $W_C5 --steps 1 --warmup 0 $COMMON

mfma() {  # name, command...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format rocpd -d $O/tmp_m_$name -- "$@" > $O/log_pmc_mfma_$name.txt 2>&1
    { echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- $* ; python tools/pmc_summary.py <db> gemm_nt_kernel"
      echo "# per-launch averages; matrix-core utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)"
      python $R/tools/pmc_summary.py $(db $O/tmp_m_$name) gemm_nt_kernel; } > $O/${TAG}_pmc_mfma_busy_$name.txt 2>&1
    rm -rf $O/tmp_m_$name
}
mfma b1 $W_B1 --steps 2 --warmup 1 $COMMON
mfma b32 $W_B32 --steps 1 --warmup 1 $COMMON
mfma b128 $W_B128 --steps 1 --warmup 0 $COMMON
mfma config3 $W_C3 --steps 1 --warmup 0 $COMMON
mfma configs3_share_1b $W_C4 --steps 1 --warmup 0 $COMMON
mfma configs4_share_1b $W_C5 --steps 1 --warmup 0 $COMMON
mfma bf16_fastmode_config3 $W_C3 --gemm bf16 --steps 1 --warmup 0 $COMMON
# per-shape GEMM time inside the model (event-timed, eager): where the image's milliseconds go
cd $R
python tools/gemm_by_shape.py 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_b1.txt
python tools/gemm_by_shape.py --batch 32 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_b32.txt
python tools/gemm_by_shape.py --batch 128 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_b128.txt
python tools/gemm_by_shape.py --model 1b --s-byt5 256 --clip-image 1 --batch 32 --grid 64 --sample-steps 2 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_configs3_share_1b_2steps.txt
python tools/gemm_by_shape.py --model 1b --s-byt5 256 --clip-image 1 --batch 16 --grid 128 --sample-steps 2 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_configs4_share_1b_2steps.txt
python tools/gemm_by_shape.py --batch 64 --grid 64 --sample-steps 2 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_config3_2steps.txt
python tools/gemm_by_shape.py --gemm bf16 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_bf16_b1.txt
python tools/gemm_by_shape.py --gemm bf16 --batch 32 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_bf16_b32.txt
python tools/gemm_by_shape.py --gemm bf16 --batch 64 --grid 64 --sample-steps 2 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_by_shape_bf16_config3_2steps.txt
# the parity report: every oracle / reference comparison with its near-tie counts printed (-s)
{ echo "# python -m pytest tests -m gpu -q -s -k 'parity or vs_oracle or vs_reference or closed_loop or benchmarked or geometry or train_step or prompts_to_image or graph_sampler or reproduces or grn_finished or nonsquare or largest_key or fused_attention or graph_inpainter'   (MI355X)"
  python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "parity or vs_oracle or vs_reference or closed_loop or benchmarked or geometry or train_step or prompts_to_image or graph_sampler or reproduces or grn_finished or nonsquare or largest_key or fused_attention or graph_inpainter" 2>&1 | grep -v "amdgpu.ids" | grep -v "^\s*$"; } > $O/${TAG}_parity_report.txt
{ echo "# python -m pytest tests/test_gpu_fastmode.py tests/test_gpu_unet.py -q -s -k 'forward_deviation or sampling_fused or vqgan or layernorm_guard'   (MI355X): bf16 fast mode deviation / flip rates; LayerNorm guard inside the network"
  python -m pytest tests/test_gpu_fastmode.py tests/test_gpu_unet.py -q -s -p no:cacheprovider -k "forward_deviation or sampling_fused or vqgan or layernorm_guard" 2>&1 | grep "fast mode\|guard in the network\|LayerNorm-folding\|sampled tokens\|passed\|failed" | sed 's/^[.F]*//'; } > $O/${TAG}_fastmode_and_ln_guard_report.txt
# round 4: the LayerNorm-fold error curve (threshold hook at inf / 0 / default), the RCCL path at the box's world size
{ echo "# python -m pytest tests/test_gpu_ops.py -q -s -k layernorm_fold   (MI355X): max |out - fp64| of a LayerNorm-consuming GEMM, K = 1280, outputs of unit scale, per |row mean| / std"
  python -m pytest tests/test_gpu_ops.py -q -s -p no:cacheprovider -k "layernorm_fold" 2>&1 | grep "cfg\|passed\|failed" | sed 's/^[.F]*//'; } > $O/${TAG}_ln_fold_error_curve.txt
bash $R/tools/two_rank_rehearsal.sh $TAG > /dev/null 2>&1; cp $R/gpurun_out/${TAG}_two_rank_rehearsal.txt $O/ 2>/dev/null
{ echo "# python -m pytest tests/test_gpu_dist.py -q -s   (MI355X, world size = GPUs of the box): torch.distributed over nccl (= RCCL) with a live process group"
  python -m pytest tests/test_gpu_dist.py -q -s -p no:cacheprovider 2>&1 | grep "^[.F]*{\|passed\|failed" | sed 's/^[.F]*//'; } > $O/${TAG}_rccl_world1.txt
# the headline line with in-date traffic: the traffic files of THIS run go where bench.py looks for them
cp $O/${TAG}_pmc_traffic_*.json $R/profiles/ 2>/dev/null
python bench.py 2> $O/log_bench_line.txt | tail -1 > $O/${TAG}_bench_line.json
ls -la $O
