#!/bin/bash
# Serialised-branch trace (POSE_MARKERS=1): per-region / per-branch kernel time -> critical-path estimate (scripts/trace_sections.py)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_sec
export POSE_MARKERS=1
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sec -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof_sec.log 2>&1 || { tail -5 gpurun_out/prof_sec.log; exit 1; }
tr=$(find gpurun_out/prof_sec -name "*kernel_trace.csv" | head -1)
python scripts/trace_sections.py "$tr" > gpurun_out/r04_trace_sections.txt 2>&1 || true
head -60 gpurun_out/r04_trace_sections.txt
rm -rf gpurun_out/prof_sec
