#!/bin/bash
# round 4, call f: A/B of which block halves run fused at C = 64 (same box, alternating)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for v in base mlp32 attn64 base2 mlp32b attn64b; do
  case $v in
    base|base2) env_="";;
    mlp32|mlp32b) env_="POSE_FUSED_MLP=32";;
    attn64|attn64b) env_="POSE_FUSED_ATTN=32,64";;
  esac
  env $env_ timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04f_$v.json 2> gpurun_out/r04f_$v.err || { tail -5 gpurun_out/r04f_$v.err | cut -c1-300; }
  python scripts/bench_ms.py gpurun_out/r04f_$v.json
done
