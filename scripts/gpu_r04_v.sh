#!/bin/bash
# round 4, call v: library built with -mllvm -amdgpu-mfma-vgpr-form=1 -- parity (network ops + parity files), then A/B on cfg 2 / 4 / 5 (B = previous build)
set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)]"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r04v_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04v_tests.log | cut -c1-300 | tail -5
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/gpu_ab_many.sh 4
bash scripts/gpu_ab_many.sh 2 hrnet_w32_384
bash scripts/gpu_ab_many.sh 2 hrformer_base_infer
