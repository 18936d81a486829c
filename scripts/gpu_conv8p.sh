#!/bin/bash
# k_conv8p: parity tests, then the head-conv micro-benchmark with the kernel on and off (same process order, same box)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 300 -k "conv" > gpurun_out/c8_tests.log 2>&1; rc=$?
tail -5 gpurun_out/c8_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E  " gpurun_out/c8_tests.log | head -20; exit $rc; fi
for v in 1 0 1 0; do
  PK_CONV8P=$v timeout -k 10 200 python scripts/bench_kernels.py "conv 256->256" 2>&1 | grep -v Warn | sed "s/^/conv8p=$v  /"
done
