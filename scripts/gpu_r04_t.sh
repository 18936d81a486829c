#!/bin/bash
# round 4, call t: spill-free k_win_attn_bwd / k_attn_bwd<32> (LayerNorm statistics and rows parked in LDS) -- parity, then A/B (B = previous build)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "attention or hrformer_block or small_train_step_vs_golden or graph_replay_matches or no_rpe or expected_gradient" > gpurun_out/r04t_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04t_tests.log | cut -c1-300 | tail -5
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/gpu_ab_many.sh 5
