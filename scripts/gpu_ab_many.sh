#!/bin/bash
# A/B of two library builds on one box, many alternations: default libposekernels.so (A) vs csrc/libposekernels_b.so (B)
# usage: bash scripts/gpu_ab_many.sh [pairs=5] [config=hrformer_small]
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
B="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc/libposekernels_b.so"
n=${1:-5}; cfg=${2:-hrformer_small}
for k in $(seq 1 $n); do
  for v in A B; do
    if [ $v = B ]; then export POSE_KERNELS_LIB="$B"; else unset POSE_KERNELS_LIB; fi
    timeout -k 10 300 python bench.py --config $cfg --steps 80 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || tail -3 gpurun_out/ab_$v.err
    echo -n "lib=$v "; python scripts/bench_ms.py gpurun_out/ab_$v.json
  done
done
