#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 200 > gpurun_out/tests_final.log 2>&1; rc=$?
tail -2 gpurun_out/tests_final.log | cut -c1-250
if [ $rc -ne 0 ]; then grep -n "^E  \|^FAILED" gpurun_out/tests_final.log | head -20; exit $rc; fi
timeout -k 10 400 python bench.py --config hrformer_base_infer --no-cpu-baseline > gpurun_out/bench_base_r02.json 2> gpurun_out/bench_base_r02.err || exit 1
timeout -k 10 400 python bench.py --config hrnet_w32_384 --no-cpu-baseline > gpurun_out/bench_w32_r02.json 2> gpurun_out/bench_w32_r02.err || exit 1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_base_r02.json gpurun_out/bench_w32_r02.json
