#!/bin/bash
# round 4, call b: grouped exchange unit tests + public surface, then bench A/B (grouped on / off)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_public_surface.py tests/test_gpu_parity.py tests/test_input_pipeline.py -m gpu -q -x --timeout 500 \
   -k "grouped or exchange or prefetch or public or constraint or term_methods or softargmax or window_partition or drop_path or w48 or without_relative or graph_replay or branch_streams or fusion_loss or train_step_vs_golden or modules_vs_golden" \
   > gpurun_out/r04b_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04b_tests.log | cut -c1-700 | tail -30
if [ $rc -ge 124 ]; then exit $rc; fi
for v in g1 g0 g2 g1b g0b; do
  case $v in
    g1|g1b) env_="POSE_GROUPED_EXCHANGE=1";;
    g2) env_="POSE_GROUPED_EXCHANGE=2";;
    g0|g0b) env_="POSE_GROUPED_EXCHANGE=0";;
  esac
  env $env_ timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04b_bench_$v.json 2> gpurun_out/r04b_bench_$v.err || { tail -8 gpurun_out/r04b_bench_$v.err | cut -c1-400; exit 1; }
  python scripts/bench_ms.py gpurun_out/r04b_bench_$v.json
done
exit $rc
