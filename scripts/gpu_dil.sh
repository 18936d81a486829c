#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "exchange or modules or fuse or upsample or eval_forward" 2>&1 | tail -2
for v in 1 2; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done
