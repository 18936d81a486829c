#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)]"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s --timeout 300 -k "rccl_gradient_exchange_and_graph" > gpurun_out/r04l_tests.log 2>&1; rc=$?
grep -E "one-rank|passed|failed|FAILED|^E  |Error|error" gpurun_out/r04l_tests.log | cut -c1-400 | tail -15
exit $rc
