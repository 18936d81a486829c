#!/bin/bash
# round 4: smoke + the full -m gpu suite with per-test durations
set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)]"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04_smoke.log | cut -c1-200
start=$(date +%s)
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 --durations=12 > gpurun_out/r04_full_tests.log 2>&1; rc=$?
echo "pytest rc=$rc in $(( $(date +%s) - start )) s"
grep -E "passed|failed|FAILED|^E  |s call|s setup" gpurun_out/r04_full_tests.log | cut -c1-300 | tail -24
exit $rc
