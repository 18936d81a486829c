#!/bin/bash
# round 4, call e: quick tests of the changed paths, sections profile (serialised branches), bench A/B of small switches
set -o pipefail
mkdir -p gpurun_out
# heartbeat: long CPU-oracle tests write nothing for minutes; gpurun kills a run that is silent for 7 minutes
( while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)]"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_public_surface.py tests/test_gpu_network_ops.py -m gpu -q -x --timeout 500 \
   -k "fusion_loss or term_methods or constraint or trainer or graph_replay or grouped or head_out or train_step_vs_golden or flip_inference or deferred" > gpurun_out/r04e_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04e_tests.log | cut -c1-600 | tail -8
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/gpu_sections.sh > gpurun_out/r04e_sections.log 2>&1; head -30 gpurun_out/r04_trace_sections.txt
for v in a b; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04e_small_$v.json 2> gpurun_out/r04e_small_$v.err || tail -3 gpurun_out/r04e_small_$v.err
done
timeout -k 10 300 python bench.py --config hrformer_base_infer --steps 30 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/r04e_base.json 2> gpurun_out/r04e_base.err || tail -3 gpurun_out/r04e_base.err
python scripts/bench_ms.py gpurun_out/r04e_small_a.json gpurun_out/r04e_small_b.json gpurun_out/r04e_base.json
