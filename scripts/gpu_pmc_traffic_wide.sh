#!/bin/bash
# L2-miss traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the wide fused halves at their cfg-5 shapes
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcw_fetch gpurun_out/pmcw_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcw_fetch -- python3 scripts/prof_kernels.py wide > gpurun_out/pmcw_fetch.log 2>&1 || { tail -5 gpurun_out/pmcw_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_write -- python3 scripts/prof_kernels.py wide > gpurun_out/pmcw_write.log 2>&1 || { tail -5 gpurun_out/pmcw_write.log; exit 1; }
python scripts/pmc_summary.py $(find gpurun_out/pmcw_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmcw_write -name "*counter_collection.csv" | head -1) > gpurun_out/r03_pmc_traffic_wide.json
rm -rf gpurun_out/pmcw_fetch gpurun_out/pmcw_write
grep -A5 "_w<" gpurun_out/r03_pmc_traffic_wide.json | head -40
