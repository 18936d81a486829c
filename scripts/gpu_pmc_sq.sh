#!/bin/bash
# SQ instruction-mix / stall counters of every kernel of one eager training step (two passes of 8 SQ counters)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sq1 gpurun_out/sq2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d gpurun_out/sq1 -- python3 scripts/prof_kernels.py step > gpurun_out/sq1.log 2>&1 || { tail -5 gpurun_out/sq1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/sq2 -- python3 scripts/prof_kernels.py step > gpurun_out/sq2.log 2>&1 || { tail -5 gpurun_out/sq2.log; exit 1; }
python scripts/pmc_sq_summary.py $(find gpurun_out/sq1 gpurun_out/sq2 -name "*counter_collection.csv") > gpurun_out/r03_pmc_sq_counters.json
rm -rf gpurun_out/sq1 gpurun_out/sq2
python - <<'PY'
import json
k=json.load(open("gpurun_out/r03_pmc_sq_counters.json"))["kernels"]
for n in ("k_attn_bwd<32>","k_attn_fwd<32>","k_mlp_fwd<32>","k_mlp_bwd_dx<32>","k_mlp_bwd_dw<32, 64>","k_mlp_bwd_dx<64>","k_conv8p","k_wgrad3","k_win_attn_bwd<1, 2>"):
    e=k.get(n,{}); print(n, {x:e[x] for x in e if x not in ("counters",)})
PY
