#!/bin/bash
# round 4, call z: grid sizes of the fused block kernels (TUNING build of the library) -- a smaller footprint of one branch's kernels
# leaves room on every CU for the other branches' workgroups.  One knob at a time against the defaults (PK_ATTN_WGS = 256 is in).
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export POSE_KERNELS_LIB="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc/libposekernels_tuning.so"
for k in 1 2; do
  for v in "PK_NONE=0" "PK_MLP_FWD32_WGS=256" "PK_MLP_FWD64_WGS=256" "PK_MLP_DX32_WGS=256" "PK_MLP_DX64_WGS=256" "PK_MLP_DW32_WGS=128" "PK_MLP_DW64_WGS=32" "PK_MLP_DX64_WGS=192" "PK_MLP_DX32_WGS=128"; do
    env $v timeout -k 10 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04z.json 2> gpurun_out/r04z.err || tail -3 gpurun_out/r04z.err
    echo -n "$v "; python scripts/bench_ms.py gpurun_out/r04z.json
  done
done
