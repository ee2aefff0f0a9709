#!/bin/bash
# Same-box A/B of the cache policy of the ring tiles' WEIGHT stream: the product library (default policy) against a probe build of
# gemm.hip with the OTHER policy (-DPAELLA_RING_W_AUX=0 since nt became the default in round 5; the committed profile was measured with 0 as the default and 2 as
# the probe) linked into a second library; the two are swapped in place and the batch-1 headline is timed alternately.
# Usage (GPU box, repo root): bash tools/ab_nt_weights.sh > gpurun_out/ring_nt_weights_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/paella_amd/csrc
T=/tmp/nt_ab; mkdir -p $T
cp $C/libpaella_hip.so $T/lib_default.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJS=""
for u in gemm elementwise dwconv attention tail vqgan model vqmodel; do
  if [ $u = gemm ]; then /opt/rocm/bin/hipcc $FLAGS -DPAELLA_RING_W_AUX=2 -c $C/gemm.hip -o $T/gemm.o 2>/dev/null
  elif [ $u = attention ]; then /opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 -c $C/$u.hip -o $T/$u.o 2>/dev/null &
  else /opt/rocm/bin/hipcc $FLAGS -c $C/$u.hip -o $T/$u.o 2>/dev/null & fi
  OBJS="$OBJS $T/$u.o"
done
wait
STAMP=$(python -c "import sys; sys.path.insert(0,'$R'); from paella_amd._stamp import source_stamp; print(source_stamp())")
printf 'static const char kStamp[] = "PAELLA_SOURCE_STAMP=%s";\nconst char* paella_source_stamp(void) { return kStamp + 20; }\n' $STAMP > $T/stamp.c
/opt/rocm/bin/hipcc -x c -O2 -fPIC -c $T/stamp.c -o $T/stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/lib_nt.so $OBJS $T/stamp.o || exit 1
echo "# ring-tile weight stream: default cache policy vs nt (aux = 2); python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline (batch 1, fp32, graph replay), alternating on one box"
for rep in 1 2 3; do
  for v in default nt; do
    cp $T/lib_$v.so $C/libpaella_hip.so
    ms=$(cd $R && python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep  $v  ms_per_image $ms"
  done
done
for v in default nt; do
    cp $T/lib_$v.so $C/libpaella_hip.so
    ms=$(cd $R && python bench.py --gemm bf16 --steps 20 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "bf16 fast mode  $v  ms_per_image $ms"
done
cp $T/lib_default.so $C/libpaella_hip.so
