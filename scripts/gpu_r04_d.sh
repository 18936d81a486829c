#!/bin/bash
# round 4, call d: fixed test, slab report, A/B of the grouped exchange on cfg 4 / cfg 5
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 250 -k "without_relative" > gpurun_out/r04d_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04d_tests.log | cut -c1-600 | tail -8
timeout -k 10 200 python scripts/slab_report.py > gpurun_out/r04d_slabs.txt 2>&1; tail -28 gpurun_out/r04d_slabs.txt
for cfgname in hrformer_base_infer hrnet_w32_384; do
for v in 1 0 1 0; do
  POSE_GROUPED_EXCHANGE=$v timeout -k 10 300 python bench.py --config $cfgname --steps 30 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/r04d_${cfgname}_$v.json 2> gpurun_out/r04d_${cfgname}_$v.err || { tail -8 gpurun_out/r04d_${cfgname}_$v.err | cut -c1-400; exit 1; }
  python scripts/bench_ms.py gpurun_out/r04d_${cfgname}_$v.json
done
done
exit $rc
