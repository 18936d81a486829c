#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/proflr
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proflr -- python3 scripts/bench_lowres_conv.py > gpurun_out/proflr.log 2>&1 || { tail -5 gpurun_out/proflr.log; exit 1; }
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/proflr/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["MinNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1))
PY
rm -rf gpurun_out/proflr
