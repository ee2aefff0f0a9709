#!/bin/bash
# round 4, GPU call 16: pricing the depthwise-conv + LayerNorm fusion into the producing GEMM's epilogue from its two measurable ingredients, per ResBlock at the 64-position level
# (M = 128 rows with CFG): (1) the out-projection / MLP-2 GEMM on an M-covering 64-row tile (33: 64x32) instead of the 32x32 ring tile (30); (2) the MLP-1 GEMM as a LayerNorm-consuming launch
# (statistics derived per workgroup, 4-stage tile 31) instead of a plain one -- against the 5.4 us dwconv_ln launch the fusion would remove.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c16
mkdir -p $O
cd $R
echo "== producer on an M-covering tile: plain operands"; timeout 300 python tools/gemm_tune.py --cfgs 30,33,21 --only "L1 out 128x1280x1280,L2 out 32x1280x1280" 2>&1 | grep "best\|ring tiles" | cut -c1-260
echo "== producer (MLP-2) with the GRN prologue"; timeout 300 python tools/gemm_tune.py --cfgs 30,33 --apro 1 --only "L1 mlp2 128x1280x5120" 2>&1 | grep "best\|ring tiles" | cut -c1-260
echo "== consumer (MLP-1) plain vs LayerNorm-consuming"; timeout 300 python tools/gemm_tune.py --cfgs 30,31 --only "L1 mlp1 128x5120x1280" 2>&1 | grep "best\|ring tiles" | cut -c1-260
timeout 300 python tools/gemm_tune.py --cfgs 30,31 --apro 2 --only "L1 mlp1 128x5120x1280" 2>&1 | grep "best\|ring tiles" | cut -c1-260
