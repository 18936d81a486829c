#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 200 -k "eval or infer or pose_inference or freshness or weights or bn or conv_bn" > gpurun_out/t_ev.log 2>&1; rc=$?
tail -2 gpurun_out/t_ev.log | cut -c1-250
if [ $rc -ne 0 ]; then grep -n "^E  \|^FAILED" gpurun_out/t_ev.log | head -20; exit $rc; fi
timeout -k 10 400 python bench.py --config hrformer_base_infer --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_base.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launch": "[a-zA-Z ]*"' | tr '\n' ' '
POSE_GRAPH=0 timeout -k 10 400 python bench.py --config hrformer_base_infer --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_base.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launch": "[a-zA-Z ]*"' | tr '\n' ' '
