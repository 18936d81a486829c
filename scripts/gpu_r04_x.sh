#!/bin/bash
# round 4, call x: library A/B with parity first (B = previous build) and the isolated time of k_attn_bwd<32> through the bench probe
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "attention or hrformer_block or small_train_step_vs_golden or graph_replay or no_rpe or cfg2_full" > gpurun_out/r04x_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04x_tests.log | cut -c1-300 | tail -5
if [ $rc -ne 0 ]; then exit $rc; fi
B="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc/libposekernels_b.so"
for v in A B; do
  if [ $v = B ]; then export POSE_KERNELS_LIB="$B"; else unset POSE_KERNELS_LIB; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2> gpurun_out/r04x_roof_$v.err
  echo "lib=$v $(grep -o 'k_attn_bwd<32>  *[0-9.]* us isolated' gpurun_out/r04x_roof_$v.err)"
done
unset POSE_KERNELS_LIB
bash scripts/gpu_ab_many.sh 4
