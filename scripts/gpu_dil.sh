#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 300 -k "conv or exchange or modules" 2>&1 | tail -2
for v in 1 0; do
  for f in "k3 s2"; do
  PK_IGEMM_DILGROUP=$v timeout -k 10 200 python scripts/bench_kernels.py "$f" 2>&1 | grep "dgrad" | sed "s/^/dilgroup=$v  /"
  done
done
for v in 1 0 1 0; do
  PK_IGEMM_DILGROUP=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/cfg2 dilgroup=$v  /"; echo
done
