#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -X faulthandler bench.py --steps 20 --warmup 5 --graph --no-cpu-baseline > gpurun_out/bench4.log 2> gpurun_out/bench4.err; rc=$?
tail -2 gpurun_out/bench4.log | cut -c1-500; tail -30 gpurun_out/bench4.err | cut -c1-250
exit 0
