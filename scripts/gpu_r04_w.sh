#!/bin/bash
# round 4, call w: k_mlp_bwd_dw<32, 128> (one hidden slice for C = 32: 255 registers with VGPR-form MFMAs) against two slices of 64 -- parity, A/B
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "mlp or hrformer_block or small_train_step_vs_golden or graph_replay_matches or expected_gradient" > gpurun_out/r04w_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04w_tests.log | cut -c1-300 | tail -5
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/gpu_ab_many.sh 4
