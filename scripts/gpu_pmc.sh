#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc -- python scripts/bench_kernels.py "256->256 k3" > gpurun_out/pmc.log 2>&1
tail -3 gpurun_out/pmc.log
find gpurun_out/pmc -name "*counter_collection.csv" | head -2
