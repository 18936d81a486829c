#!/bin/bash
# SQ counters of the low-resolution deep-K conv launches (scripts/bench_lowres_conv.py)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sql1 gpurun_out/sql2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d gpurun_out/sql1 -- python3 scripts/bench_lowres_conv.py > gpurun_out/sql1.log 2>&1 || { tail -5 gpurun_out/sql1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/sql2 -- python3 scripts/bench_lowres_conv.py > gpurun_out/sql2.log 2>&1 || { tail -5 gpurun_out/sql2.log; exit 1; }
python scripts/pmc_sq_summary.py $(find gpurun_out/sql1 gpurun_out/sql2 -name "*counter_collection.csv") > gpurun_out/r03_pmc_sq_lowres.json
rm -rf gpurun_out/sql1 gpurun_out/sql2
python - <<'PY'
import json
k=json.load(open("gpurun_out/r03_pmc_sq_lowres.json"))["kernels"]
for n,e in k.items():
    if "igemm" in n: print(n, {x:(e[x] if x!="counters" else {c:e[x][c] for c in ("SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_INSTS_MFMA","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS") if c in e[x]}) for x in e})
PY
