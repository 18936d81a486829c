#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 300 -k "conv" 2>&1 | tail -2
for v in 1 0 1 0; do PK_IGEMM_CHUNK_MAJOR=$v timeout -k 10 200 python scripts/bench_kernels.py "conv 256->32" 2>&1 | grep "fwd\|dgrad" | sed "s/^/cm=$v /"; done
