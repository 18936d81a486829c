#!/bin/bash
# round 4, call o: k_attn_bwd accumulates du from its own packed dq / dk / dv fragments (no dqkv read-back) -- parity, then A/B (B = previous build)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "attention or hrformer_block or small_train_step_vs_golden or graph_replay_matches or no_rpe or expected_gradient" > gpurun_out/r04o_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04o_tests.log | cut -c1-400 | tail -6
if [ $rc -ne 0 ]; then exit $rc; fi
B="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc/libposekernels_b.so"
for v in A B A B; do
  if [ $v = B ]; then export POSE_KERNELS_LIB="$B"; else unset POSE_KERNELS_LIB; fi
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04o_$v.json 2> gpurun_out/r04o_$v.err || tail -3 gpurun_out/r04o_$v.err
  echo "lib=$v"; python scripts/bench_ms.py gpurun_out/r04o_$v.json
done
unset POSE_KERNELS_LIB
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04o_roof.json 2> gpurun_out/r04o_roof.err
grep -i "attn_bwd" gpurun_out/r04o_roof.err | head -5
