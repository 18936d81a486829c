#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for e in "X=1" "PK_IGEMM_SHALLOW32=64" "PK_IGEMM_SHALLOW32=64 PK_IGEMM_STATS64=1" "PK_IGEMM_STATS64=1" "PK_IGEMM_SHALLOW32=256 PK_IGEMM_STATS64=1"; do
  for f in "conv 64->256 k1" "conv 256->64 k1"; do
    env $e timeout -k 10 300 python scripts/bench_kernels.py "$f" 2>&1 | grep "fwd\|dgrad" | sed "s/^/[$e] /"
  done
done
