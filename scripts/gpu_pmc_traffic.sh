#!/bin/bash
# HBM traffic of the dominant kernel (head conv 3x3 256->256 fwd) from the TCC fabric-side counters, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; no trace domains besides the implicit kernel dispatch records).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python scripts/bench_kernels.py "256->256 k3" > gpurun_out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python scripts/bench_kernels.py "256->256 k3" > gpurun_out/pmc_write.log 2>&1 || exit 1
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv"
