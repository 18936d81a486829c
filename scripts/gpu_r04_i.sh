#!/bin/bash
# round 4, call i: which tasks of a fork / join share a stream (POSE_STREAM_MAP), same box, alternating; + the fixed drop_path test
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python -m pytest tests/test_gpu_public_surface.py -m gpu -q -x --timeout 150 -k "drop_path" 2>&1 | tail -1
for v in base m0012 m0011 m0122 m0112 base2 m0012b; do
  case $v in
    base|base2) env_="";;
    m0012|m0012b) env_="POSE_STREAM_MAP=0,0,1,2";;
    m0011) env_="POSE_STREAM_MAP=0,0,1,1";;
    m0122) env_="POSE_STREAM_MAP=0,1,2,2";;
    m0112) env_="POSE_STREAM_MAP=0,1,1,2";;
  esac
  env $env_ timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04i_$v.json 2> gpurun_out/r04i_$v.err || { tail -5 gpurun_out/r04i_$v.err | cut -c1-300; }
  python scripts/bench_ms.py gpurun_out/r04i_$v.json
done
