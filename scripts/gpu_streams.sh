#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/tests_all.log 2>&1; tail -4 gpurun_out/tests_all.log | cut -c1-250
for st in 0 1; do
  echo "== POSE_STREAMS=$st eager"; POSE_STREAMS=$st timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'
  echo "== POSE_STREAMS=$st graph"; POSE_STREAMS=$st timeout -k 10 300 python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/streams_$st.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*'
  tail -3 gpurun_out/streams_$st.err | cut -c1-200
done
