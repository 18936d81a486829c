#!/bin/bash
# round 4, call y: public-surface tests (incl. the empty-batch test)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_public_surface.py -m gpu -q -x --timeout 300 > gpurun_out/r04y_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04y_tests.log | cut -c1-300 | tail -8
exit $rc
