#!/bin/bash
# Full GPU check of the current tree: all gpu tests, smoke, a rocprofv3 kernel trace of the default bench, and
# un-profiled bench lines (default = hipGraph + branch streams, then eager for comparison).
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/round_tests.log 2>&1; rc=$?
tail -4 gpurun_out/round_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/round_tests.log | head -20; exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline > gpurun_out/prof.log 2>&1 || exit 1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/prof.log
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_default.json || exit 1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launch": "[a-zA-Z, ]*"\|"roofline": {[^}]*}' gpurun_out/bench_default.json
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --eager 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launch": "[a-zA-Z, ]*"'
