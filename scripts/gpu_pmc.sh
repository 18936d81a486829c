#!/bin/bash
# SQ counters of the kernels a bench_kernels.py filter launches:  bash scripts/gpu_pmc.sh "<filter>" <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
flt=${1:-"256->256 k3"}; tag=${2:-pmc}
rm -rf gpurun_out/$tag
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/$tag -- python scripts/bench_kernels.py "$flt" > gpurun_out/$tag.log 2>&1
tail -3 gpurun_out/$tag.log
f=$(find gpurun_out/$tag -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
for k, c in agg.items():
    w = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:50s} n={n[k]:4d} wave_cyc/launch={w / max(n[k], 1):12.0f} wait_any={c['SQ_WAIT_ANY'] / w:.2f} wait_inst={c['SQ_WAIT_INST_ANY'] / w:.2f} "
          f"active={c['SQ_ACTIVE_INST_ANY'] / w:.2f} valu_active={c['SQ_ACTIVE_INST_VALU'] / w:.2f} insts_valu/launch={c['SQ_INSTS_VALU'] / max(n[k], 1):.0f} "
          f"lds_active={c['SQ_LDS_IDX_ACTIVE'] / w:.3f} lds_conf={c['SQ_LDS_BANK_CONFLICT'] / w:.3f}")
PY
rm -rf gpurun_out/$tag
