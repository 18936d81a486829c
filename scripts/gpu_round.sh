#!/bin/bash
# Full GPU check of the current tree: all gpu tests, smoke, a rocprofv3 kernel trace of the default bench, and
# an un-profiled bench line.  Usage (from the repo root, through gpurun): bash scripts/gpu_round.sh <tag>
set -o pipefail
tag=${1:-cur}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 200 > gpurun_out/tests_$tag.log 2>&1; rc=$?
tail -4 gpurun_out/tests_$tag.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  \|^FAILED" gpurun_out/tests_$tag.log | head -30; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || { tail -20 gpurun_out/prof_$tag.log; exit 1; }
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/prof_$tag.log
python scripts/trace_summary.py $(ls gpurun_out/prof_$tag/*/*kernel_trace.csv | head -1) 3 > gpurun_out/trace_summary_$tag.txt 2>&1
cp $(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_$tag.csv
rm -rf gpurun_out/prof_$tag
head -34 gpurun_out/trace_summary_$tag.txt; tail -28 gpurun_out/trace_summary_$tag.txt
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_$tag.json || exit 1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launch": "[a-zA-Z, ]*"' gpurun_out/bench_$tag.json
