#!/bin/bash
# round 4, first GPU call: new public-surface tests + no-RPE graph test, then bench A/B (detached last exchange, serial head branches)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_public_surface.py tests/test_gpu_parity.py -m gpu -q -x --timeout 500 \
   -k "public or constraint or term_methods or softargmax or window_partition or drop_path or w48 or without_relative or graph_replay or branch_streams or fusion_loss" \
   > gpurun_out/r04a_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04a_tests.log | cut -c1-600 | tail -30
if [ $rc -ge 124 ]; then exit $rc; fi
for v in base nodetach headserial base2 headserial2; do
  case $v in
    base|base2) env_="";;
    nodetach) env_="POSE_LAST_EXCHANGE_DETACHED=0";;
    headserial|headserial2) env_="POSE_HEAD_SERIAL=1";;
  esac
  env $env_ timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04a_bench_$v.json 2> gpurun_out/r04a_bench_$v.err || { tail -5 gpurun_out/r04a_bench_$v.err; exit 1; }
  echo "$v: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r04a_bench_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
done
exit $rc
