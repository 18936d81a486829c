#!/bin/bash
# round 4, call k: which widths take the fused MLP half in TRAINING, re-measured on the final code (same box, alternating)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for v in a32 b3264 c32 d3264 e32 f3264; do
  case $v in
    a32|c32|e32) env_="POSE_FUSED_MLP=32";;
    *) env_="POSE_FUSED_MLP=32,64";;
  esac
  env $env_ timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04k_$v.json 2> gpurun_out/r04k_$v.err || tail -3 gpurun_out/r04k_$v.err
  python scripts/bench_ms.py gpurun_out/r04k_$v.json
done
