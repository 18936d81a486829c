#!/bin/bash
# round 4, call z: grid sizes on the TUNING build -- persistent workgroups per CU of the halo conv kernel (cfg 4's branches 0 / 1 are chains of it)
# (the TUNING library: copy csrc/ + include/ to a scratch directory, `make TUNING=1` there, copy its libposekernels.so to
#  infantposeestimation_gaussianbias_amd/csrc/libposekernels_tuning.so before the gpurun call; the release library ignores these knobs)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export POSE_KERNELS_LIB="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc/libposekernels_tuning.so"
for k in 1 2 3; do
  for v in "PK_CONV3H_PER_CU=2" "PK_CONV3H_PER_CU=1" "PK_CONV3H_PER_CU=3"; do
    env $v timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 40 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r04z.json 2> gpurun_out/r04z.err || tail -3 gpurun_out/r04z.err
    echo -n "cfg4 $v "; python scripts/bench_ms.py gpurun_out/r04z.json
  done
done
for v in "PK_CONV3H_PER_CU=2" "PK_CONV3H_PER_CU=1"; do
  env $v timeout -k 10 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r04z.json 2> gpurun_out/r04z.err || tail -3 gpurun_out/r04z.err
  echo -n "cfg2 $v "; python scripts/bench_ms.py gpurun_out/r04z.json
done
