#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/prof.log
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launch": "[a-zA-Z ]*"'
