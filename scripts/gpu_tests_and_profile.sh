#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -s --timeout 300 > gpurun_out/tests3.log 2>&1; rc=$?
grep -E "max-norm|^l2|passed|failed|FAILED" gpurun_out/tests3.log | cut -c1-1500 | tail -40
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof3.log 2>&1; rc=$?
tail -2 gpurun_out/prof3.log | cut -c1-600
find gpurun_out/prof3 -name "*kernel_stats*" | head -3
exit 0
