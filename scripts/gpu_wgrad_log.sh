#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
POSE_LOG_WGRAD=1 timeout -k 10 300 python bench.py --steps 1 --warmup 3 --eager --no-cpu-baseline --no-roofline > gpurun_out/wgrad_log.txt 2>&1
grep -c "^#" gpurun_out/wgrad_log.txt
