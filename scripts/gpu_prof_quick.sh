#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof3
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof3.log 2>&1 || { tail -5 gpurun_out/prof3.log; exit 1; }
st=$(find gpurun_out/prof3 -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/prof3 -name "*kernel_trace.csv" | head -1)
cp "$st" gpurun_out/r03_bench_kernel_stats.csv
python scripts/trace_summary.py "$tr" > gpurun_out/r03_trace_summary.txt 2>&1 || true
head -45 gpurun_out/r03_trace_summary.txt
rm -rf gpurun_out/prof3
