#!/bin/bash
# round 4, call j (TUNING build of the library): 8-wave C = 64 MLP backward dx kernel -- parity tests, then A/B
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "fused_mlp or hrformer_block or small_train_step_vs_golden or graph_replay_matches" > gpurun_out/r04j_tests.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|^E  " gpurun_out/r04j_tests.log | cut -c1-400 | tail -6
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 1 0 1 0; do
  PK_MLP_DX8=$v timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04j_$v.json 2> gpurun_out/r04j_$v.err || tail -3 gpurun_out/r04j_$v.err
  python scripts/bench_ms.py gpurun_out/r04j_$v.json
done
