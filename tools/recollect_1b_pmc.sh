#!/bin/bash
# HBM traffic + matrix-core busy counters of the two 1B shares (BASELINE configs[3] / configs[4] per GPU).  rocprofv3 --pmc segfaulted on these workloads when the
# profiled command also ran bench.py's event-bracketed roofline pass (~10 k counter-collected dispatches, gpurun_out/r06_profiles/log_pmc_fetch_configs3_share_1b.txt);
# with --no-roofline (the timed steps alone) it does not.  If a pass still leaves no database, it is retried at --sample-steps 4: the per-LAUNCH averages the traffic
# file holds do not depend on the number of steps (same launch mix per step; the once-per-request launches weigh 3x more) -- the JSON's `command` records what ran.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { find $1 -name "*.db" 2>/dev/null | head -1; }
run() {  # name, counters, steps, more flags...
    local name=$1 ctr=$2 steps=$3; shift 3
    rm -rf $O/tmp_$name
    rocprofv3 --kernel-trace --pmc $ctr --output-format rocpd -d $O/tmp_$name -- python $R/bench.py --model 1b --s-byt5 256 --clip-image 1 --sample-steps $steps "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-graph --no-roofline > $O/log_$name.txt 2>&1
}
one() {  # share name, batch, grid, more flags
    local share=$1 b=$2 g=$3 more=$4
    local steps=12
    run f_$share FETCH_SIZE $steps --batch $b --grid $g $more
    if [ -z "$(db $O/tmp_f_$share)" ]; then steps=4; run f_$share FETCH_SIZE $steps --batch $b --grid $g $more; fi
    run w_$share WRITE_SIZE $steps --batch $b --grid $g $more
    python $R/tools/pmc_traffic.py $O/tmp_f_$share $O/tmp_w_$share $O/${TAG}_pmc_traffic_$share.json $b $g 12 1b "--s-byt5 256 --clip-image 1 $more --no-roofline (profiled at --sample-steps $steps)" > $O/log_pmc_traffic_$share.txt 2>&1
    run m_$share "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" $steps --batch $b --grid $g $more
    { echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python bench.py --model 1b --s-byt5 256 --clip-image 1 --batch $b --grid $g --sample-steps $steps $more --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-graph --no-roofline ; python tools/pmc_summary.py <db> gemm_nt_kernel   (and attention_lds_kernel)"
      echo "# per-launch averages; matrix-core utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)"
      python $R/tools/pmc_summary.py $(db $O/tmp_m_$share) gemm_nt_kernel; python $R/tools/pmc_summary.py $(db $O/tmp_m_$share) attention_lds_kernel; } > $O/${TAG}_pmc_mfma_busy_$share.txt 2>&1
    rm -rf $O/tmp_f_$share $O/tmp_w_$share $O/tmp_m_$share
}
one configs3_share_1b 32 64 ""
one configs4_share_1b 16 128 "--inpaint"
cp $O/${TAG}_pmc_traffic_configs*_share_1b.json $R/profiles/ 2>/dev/null
ls -la $O | grep "1b"
