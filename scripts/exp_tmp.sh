#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
rm -rf gpurun_out/prof_rm
PK_REDUCE_WIDE=$v timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rm -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline > gpurun_out/prof_rm.log 2>&1 || { tail -20 gpurun_out/prof_rm.log; exit 1; }
echo "PK_REDUCE_WIDE=$v"; grep "k_reduce_many\|k_adamw" $(ls gpurun_out/prof_rm/*/*kernel_stats.csv | head -1) | cut -c1-160
done
rm -rf gpurun_out/prof_rm
