#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 400 -k "conv_fwd" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s --timeout 400 -k "$1" > gpurun_out/one.log 2>&1
grep -E "losses HIP|heatmaps L2|gradient L2|^E  |passed|failed" gpurun_out/one.log | cut -c1-400 | head -20
