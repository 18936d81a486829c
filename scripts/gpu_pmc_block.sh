#!/bin/bash
# L2-miss traffic of the HRFormer block kernels on branch 0 (C = 32, 64x48, B = 64): fused kernels and the unfused sequences, from
# the TCC fabric-side counters, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; no trace domains).
# Usage (through gpurun, from the repo root): bash scripts/gpu_pmc_block.sh <tag>
set -o pipefail
tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcb_fetch gpurun_out/pmcb_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcb_fetch -- python scripts/bench_kernels.py "block C=32" > gpurun_out/pmcb_fetch.log 2>&1 || { tail -5 gpurun_out/pmcb_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcb_write -- python scripts/bench_kernels.py "block C=32" > gpurun_out/pmcb_write.log 2>&1 || { tail -5 gpurun_out/pmcb_write.log; exit 1; }
python scripts/pmc_summary.py $(find gpurun_out/pmcb_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmcb_write -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_block_$tag.json
rm -rf gpurun_out/pmcb_fetch gpurun_out/pmcb_write
cat gpurun_out/pmc_block_$tag.json
