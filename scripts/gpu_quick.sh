#!/bin/bash
# quick A/B: conv micro-benchmark + the default bench line, with an env assignment list in $1 (e.g. "PK_CONV8P=0")
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 300 -k "conv8p or conv_fwd" 2>&1 | tail -2
timeout -k 10 200 python scripts/bench_kernels.py "conv 256->256" 2>&1 | grep "conv 256"
for v in 1 0 1 0; do
  PK_CONV8P=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/conv8p=$v  /"; echo
done
