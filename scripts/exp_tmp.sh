#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 200 > gpurun_out/t_sk.log 2>&1; rc=$?
tail -2 gpurun_out/t_sk.log | cut -c1-250
if [ $rc -ne 0 ]; then grep -n "^E  \|^FAILED" gpurun_out/t_sk.log | head -20; exit $rc; fi
for v in 0 1 0 1; do POSE_SKIP_GRAD=$v timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>gpurun_out/sw.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' || { tail -5 gpurun_out/sw.err; exit 1; }; echo " skip=$v"; done
for v in 0 1; do POSE_SKIP_GRAD=$v timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 30 --warmup 6 --no-cpu-baseline 2>gpurun_out/sw.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' || { tail -5 gpurun_out/sw.err; exit 1; }; echo " w32 skip=$v"; done
