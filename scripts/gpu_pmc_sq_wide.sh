#!/bin/bash
# SQ instruction-mix / stall counters + L2-miss traffic of the wide fused halves at their cfg-5 shapes
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sqw1 gpurun_out/sqw2 gpurun_out/sqw3 gpurun_out/sqw4
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d gpurun_out/sqw1 -- python3 scripts/prof_kernels.py wide > gpurun_out/sqw1.log 2>&1 || { tail -5 gpurun_out/sqw1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/sqw2 -- python3 scripts/prof_kernels.py wide > gpurun_out/sqw2.log 2>&1 || { tail -5 gpurun_out/sqw2.log; exit 1; }
python scripts/pmc_sq_summary.py $(find gpurun_out/sqw1 gpurun_out/sqw2 -name "*counter_collection.csv") > gpurun_out/r03_pmc_sq_counters_wide.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sqw3 -- python3 scripts/prof_kernels.py wide > gpurun_out/sqw3.log 2>&1 || { tail -5 gpurun_out/sqw3.log; exit 1; }
cp $(find gpurun_out/sqw3 -name "*kernel_stats.csv" | head -1) gpurun_out/r03_wide_kernel_stats.csv
rm -rf gpurun_out/sqw1 gpurun_out/sqw2 gpurun_out/sqw3
python - <<'PY'
import json, csv
k=json.load(open("gpurun_out/r03_pmc_sq_counters_wide.json"))["kernels"]
for n,e in k.items():
    if "_w<" in n: print(n, {x:e[x] for x in e if x not in ("counters",)})
for r in csv.DictReader(open("gpurun_out/r03_wide_kernel_stats.csv")):
    if "_w<" in r["Name"]: print(r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3, "us")
PY
