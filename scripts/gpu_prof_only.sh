#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench (hipGraph + branch streams) and of the eager variant (per-stream ids)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_graph gpurun_out/prof_eager
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_graph -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline > gpurun_out/prof_graph.log 2>&1 || exit 1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/prof_graph.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_eager -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --eager > gpurun_out/prof_eager.log 2>&1 || exit 1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/prof_eager.log
