#!/bin/bash
# round 4, call c: full -m gpu suite, then kernel trace + summary of the default bench, then cfg 4 / cfg 5 quick lines
set -o pipefail
mkdir -p gpurun_out
# heartbeat: long CPU-oracle tests write nothing for minutes; gpurun kills a run that is silent for 7 minutes
( while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)]"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -s --timeout 600 -x > gpurun_out/r04c_tests.log 2>&1; rc=$?
grep -E "expected-gradient|resident|passed|failed|FAILED|^E  " gpurun_out/r04c_tests.log | cut -c1-800 | tail -30
if [ $rc -ge 124 ]; then exit $rc; fi
rm -rf gpurun_out/prof4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof4.log 2>&1 || { tail -5 gpurun_out/prof4.log; exit 1; }
st=$(find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/prof4 -name "*kernel_trace.csv" | head -1)
cp "$st" gpurun_out/r04c_bench_kernel_stats.csv
python scripts/trace_summary.py "$tr" > gpurun_out/r04c_trace_summary.txt 2>&1 || true
head -8 gpurun_out/r04c_trace_summary.txt; grep "in flight" gpurun_out/r04c_trace_summary.txt
rm -rf gpurun_out/prof4
timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r04c_w32.json 2> gpurun_out/r04c_w32.err || tail -3 gpurun_out/r04c_w32.err
timeout -k 10 300 python bench.py --config hrformer_base_infer --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r04c_base.json 2> gpurun_out/r04c_base.err || tail -3 gpurun_out/r04c_base.err
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04c_small.json 2> gpurun_out/r04c_small.err || tail -3 gpurun_out/r04c_small.err
python scripts/bench_ms.py gpurun_out/r04c_w32.json gpurun_out/r04c_base.json gpurun_out/r04c_small.json
exit $rc
