#!/bin/bash
set -o pipefail
bash scripts/gpu_round.sh r02d > gpurun_out/round_r02d.log 2>&1 || { tail -30 gpurun_out/round_r02d.log; exit 1; }
grep -E "passed|smoke ok|\"value\"|ms_per_step|launches/step|sum of kernel|in flight" gpurun_out/round_r02d.log | head -12
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_sec
POSE_MARKERS=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sec -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/prof_sec.log 2>&1 || { tail -20 gpurun_out/prof_sec.log; exit 1; }
python scripts/trace_sections.py $(ls gpurun_out/prof_sec/*/*kernel_trace.csv | head -1) > gpurun_out/trace_sections_r02d.txt 2>&1
rm -rf gpurun_out/prof_sec
head -3 gpurun_out/trace_sections_r02d.txt
timeout -k 10 600 python bench.py > gpurun_out/bench_line_r02d.json 2> gpurun_out/bench_line_r02d.err || { tail -5 gpurun_out/bench_line_r02d.err; exit 1; }
grep "roofline" gpurun_out/bench_line_r02d.err | cut -c1-160
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_line_r02d.json | head -2
