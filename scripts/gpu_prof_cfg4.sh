#!/bin/bash
# kernel statistics of BASELINE cfg 4 (HRNet-W32 + heatmap head, 384x288, training step)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof4
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 bench.py --config hrnet_w32_384 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof4.log 2>&1 || { tail -5 gpurun_out/prof4.log; exit 1; }
st=$(find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/prof4 -name "*kernel_trace.csv" | head -1)
cp "$st" gpurun_out/r04_cfg4_kernel_stats.csv
python scripts/trace_summary.py "$tr" > gpurun_out/r04_cfg4_trace_summary.txt 2>&1 || true
grep -E "^\{" gpurun_out/prof4.log | cut -c1-200
rm -rf gpurun_out/prof4
