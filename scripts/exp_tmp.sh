#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 120 -k "wgrad_window or rowmaps" > gpurun_out/t_w4.log 2>&1; rc=$?
tail -3 gpurun_out/t_w4.log | cut -c1-250
if [ $rc -ne 0 ]; then grep -n "^E  \|^FAILED" gpurun_out/t_w4.log | head -20; exit $rc; fi
timeout -k 10 500 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 200 > gpurun_out/t_w4b.log 2>&1; rc=$?
tail -2 gpurun_out/t_w4b.log | cut -c1-250
if [ $rc -ne 0 ]; then grep -n "^E  \|^FAILED" gpurun_out/t_w4b.log | head -20; exit $rc; fi
for v in 7 15 7 15; do PK_WGRAD4=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>gpurun_out/w4.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo " PK_WGRAD4=$v"; done
