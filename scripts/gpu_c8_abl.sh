#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for g in 256 100000 512 384 256 100000; do
  PK_CONV8P_GRID=$g timeout -k 10 200 python scripts/bench_kernels.py "conv 256->256" 2>&1 | grep "dgrad\|fwd" | sed "s/^/grid=$g  /"
done
