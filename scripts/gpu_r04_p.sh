#!/bin/bash
# round 4, call p: the xor-16 reductions through ds_swizzle instead of v_permlane16_swap -- determinism probes, parity tests, A/B
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
D="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc"
echo "=== locate2, 4-wave launch of the wide attention kernel (the most sensitive configuration)"
env POSE_KERNELS_LIB=$D/libposekernels_w4n.so PROBE_N=80 timeout -k 10 300 python scripts/probes/cfg5_locate2.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "=== locate2, release library"
env PROBE_N=80 timeout -k 10 300 python scripts/probes/cfg5_locate2.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
for v in "PROBE_MODE=graph PROBE_N=150" "PROBE_MODE=eager PROBE_N=80"; do
  echo "=== poison $v"
  env $v timeout -k 10 300 python scripts/probes/cfg5_poison.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
done
timeout -k 10 600 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "attention or hrformer_block or layernorm or mlp or small_train_step_vs_golden or graph_replay or cfg5 or loss" > gpurun_out/r04p_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04p_tests.log | cut -c1-300 | tail -5
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/gpu_ab_many.sh 4
