#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in 1 0; do
  for f in "conv 64->64 k3 s1" "conv 32->32 k3 s1" "conv 32->64 k3 s1" "conv 64->32 k3 s1"; do
  PK_CONV3H=$v PK_CONV3H_MIN_TILES=1 timeout -k 10 200 python scripts/bench_kernels.py "$f" 2>&1 | grep "fwd\|dgrad" | sed "s/^/c3h=$v  /"
  done
done
for v in 1 0 1 0; do
  PK_CONV3H=$v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/cfg2 c3h=$v  /"; echo
done
for v in 1 0; do
  PK_CONV3H=$v timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 20 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/cfg4 c3h=$v  /"; echo
done
