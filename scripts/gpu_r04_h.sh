#!/bin/bash
# round 4, call h: ReLU bit masks in BatchNorm backward -- tests of the touched paths, then A/B on cfg 2 and cfg 4
set -o pipefail
mkdir -p gpurun_out
# heartbeat: long CPU-oracle tests write nothing for minutes; gpurun kills a run that is silent for 7 minutes
( while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)]"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py tests/test_gpu_public_surface.py -m gpu -q -x --timeout 600 \
   -k "conv_bn or modules_vs_golden or exchange or grouped or eval_mode_batchnorm or train_step or trajectory or graph_replay or deconv or reproducible or deferred or w32" > gpurun_out/r04h_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04h_tests.log | cut -c1-600 | tail -10
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 1 0 1 0; do
  POSE_RELU_BITMASK=$v timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04h_small_$v.json 2> gpurun_out/r04h_small_$v.err || tail -3 gpurun_out/r04h_small_$v.err
  python scripts/bench_ms.py gpurun_out/r04h_small_$v.json
done
for v in 1 0 1 0; do
  POSE_RELU_BITMASK=$v timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 30 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/r04h_w32_$v.json 2> gpurun_out/r04h_w32_$v.err || tail -3 gpurun_out/r04h_w32_$v.err
  python scripts/bench_ms.py gpurun_out/r04h_w32_$v.json
done
