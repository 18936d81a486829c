#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 1 0 1; do
  PK_CONV8P=$v timeout -k 10 200 python scripts/bench_kernels.py "conv 256->256" 2>&1 | grep "conv 256" | sed "s/^/conv8p=$v  /"
done
rocprofv3 -L > gpurun_out/counters.txt 2>&1
grep -c . gpurun_out/counters.txt
rm -rf gpurun_out/c8_trace gpurun_out/c8_pmc1 gpurun_out/c8_pmc2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c8_trace -- python3 scripts/prof_conv.py fwd fwd_nostats dgrad > gpurun_out/c8_trace.log 2>&1
find gpurun_out/c8_trace -name "*kernel_stats.csv" -exec head -8 {} \;
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/c8_pmc1 -- python3 scripts/prof_conv.py fwd_nostats dgrad > gpurun_out/c8_pmc1.log 2>&1
tail -2 gpurun_out/c8_pmc1.log
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/c8_pmc2 -- python3 scripts/prof_conv.py fwd_nostats dgrad > gpurun_out/c8_pmc2.log 2>&1
tail -2 gpurun_out/c8_pmc2.log
ls gpurun_out/c8_pmc1 gpurun_out/c8_pmc2 | head
