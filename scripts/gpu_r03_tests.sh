#!/bin/bash
# full -m gpu suite + one default bench line (round 3 baseline)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -s --timeout 600 -x > gpurun_out/r03_tests.log 2>&1; rc=$?
grep -E "expected-gradient|largest distance|cfg 2 full size|passed|failed|FAILED|^E  " gpurun_out/r03_tests.log | cut -c1-1200 | tail -40
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; fi
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 50 --warmup 10 > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err; brc=$?
tail -3 gpurun_out/r03_bench0.err | cut -c1-300
cut -c1-400 gpurun_out/r03_bench0.json
exit $rc
