#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 200 > gpurun_out/t_w4.log 2>&1; rc=$?
tail -2 gpurun_out/t_w4.log | cut -c1-250
if [ $rc -ne 0 ]; then grep -n "^E  \|^FAILED" gpurun_out/t_w4.log | head -20; exit $rc; fi
for v in 1 2 3; do timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>gpurun_out/sw.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' || { tail -5 gpurun_out/sw.err; exit 1; }; echo; done
