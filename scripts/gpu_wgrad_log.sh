#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
PK_IGEMM_LOG=1 timeout -k 10 120 python bench.py --steps 1 --warmup 3 --eager --no-cpu-baseline --no-roofline > gpurun_out/igemm_log.txt 2>&1
grep -c "^# igemm" gpurun_out/igemm_log.txt
