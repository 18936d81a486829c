#!/bin/bash
# round 4, call n: slab launches (weight gradients) of the low-resolution branches on auxiliary streams -- parity, then A/B on cfg 4 and cfg 2
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
POSE_WGRAD_AUX=2 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "graph_replay_matches or small_train_step_vs_golden or hrnet" > gpurun_out/r04n_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04n_tests.log | cut -c1-400 | tail -6
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 0 1 2 0 1 2; do
  POSE_WGRAD_AUX=$v timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04n_c4_$v.json 2> gpurun_out/r04n_c4_$v.err || tail -3 gpurun_out/r04n_c4_$v.err
  echo "cfg4 aux=$v"; python scripts/bench_ms.py gpurun_out/r04n_c4_$v.json
done
for v in 0 2 0 2; do
  POSE_WGRAD_AUX=$v timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04n_c2_$v.json 2> gpurun_out/r04n_c2_$v.err || tail -3 gpurun_out/r04n_c2_$v.err
  echo "cfg2 aux=$v"; python scripts/bench_ms.py gpurun_out/r04n_c2_$v.json
done
