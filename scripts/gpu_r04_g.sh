#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_public_surface.py tests/test_gpu_parity.py -m gpu -q -x --timeout 500 -k "deconv or w18 or w32 or cfg1 or data_parallel_two" > gpurun_out/r04g_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04g_tests.log | cut -c1-700 | tail -12
exit $rc
