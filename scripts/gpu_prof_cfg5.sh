#!/bin/bash
# kernel statistics of BASELINE cfg 5 (HRFormer-base flip-test inference)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -- python3 bench.py --config hrformer_base_infer --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof5.log 2>&1 || { tail -5 gpurun_out/prof5.log; exit 1; }
st=$(find gpurun_out/prof5 -name "*kernel_stats.csv" | head -1)
cp "$st" gpurun_out/r04_cfg5_kernel_stats.csv
tail -2 gpurun_out/prof5.log
rm -rf gpurun_out/prof5
