#!/bin/bash
# quick iteration: network-op tests + kernel microbench
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_network_ops.py -m gpu -q --timeout 200 > gpurun_out/iter_tests.log 2>&1; rc=$?
tail -6 gpurun_out/iter_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/iter_tests.log | head -10; exit $rc; fi
timeout -k 10 300 python scripts/bench_kernels.py $1 > gpurun_out/kbench.log 2>&1
cat gpurun_out/kbench.log
