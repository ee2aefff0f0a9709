TAG=r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra --no-graph"
db() { find $1 -name "*.db" | head -1; }
trace() {
    local name=$1 images=$2; shift 2
    rocprofv3 --kernel-trace --output-format rocpd -d $O/tmp_$name -- "$@" > $O/log_trace_$name.txt 2>&1
    { echo "# rocprofv3 --kernel-trace --output-format rocpd -- $* ; python tools/prof_summary.py <db> $images   (MI355X)"; python $R/tools/prof_summary.py $(db $O/tmp_$name) $images; } > $O/${TAG}_kernel_trace_$name.txt 2>&1
    rm -rf $O/tmp_$name
}
W_C4="python $R/bench.py --model 1b --batch 32 --grid 64 --sample-steps 12 --s-byt5 256 --clip-image 1"
W_C5="python $R/bench.py --model 1b --batch 16 --grid 128 --sample-steps 12 --s-byt5 256 --clip-image 1 --inpaint"
trace configs3_share_1b_b32_64x64 64 $W_C4 --steps 1 --warmup 0 $COMMON
trace configs4_share_1b_b16_128x128_inpaint 32 $W_C5 --steps 1 --warmup 0 $COMMON
cd $R
python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids > $O/attn_probe_final.txt
ls -la $O
