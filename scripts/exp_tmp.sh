#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for v in 0 1 0 1; do POSE_EXCHANGE_ROUTES=$v timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>gpurun_out/sw.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' || { tail -5 gpurun_out/sw.err; exit 1; }; echo " routes=$v"; done
