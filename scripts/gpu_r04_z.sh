#!/bin/bash
# round 4, call z: grid sizes on the TUNING build -- workgroup targets of the weight-gradient kernels (fewer slices = smaller footprint and fewer slab bytes)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export POSE_KERNELS_LIB="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc/libposekernels_tuning.so"
for k in 1 2 3; do
  for v in "PK_NONE=0" "PK_WGRAD4_WGS=256" "PK_WGRAD4_WGS=384" "PK_WGRAD4_WGS9=128" "PK_WGRAD4_WGS=256 PK_WGRAD4_WGS9=128"; do
    env $v timeout -k 10 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04z.json 2> gpurun_out/r04z.err || tail -3 gpurun_out/r04z.err
    echo -n "$v "; python scripts/bench_ms.py gpurun_out/r04z.json
  done
done
