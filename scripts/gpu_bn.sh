#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 400 -k "conv_bn or modules or exchange or stem or train_step or trainer_two or bitwise or region_mode or graph_replay or w32 or cfg1" 2>&1 | tail -3
for v in 1 0 1 0; do
  PK_BN_FUSED=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/bn_fused=$v  /"; echo
done
