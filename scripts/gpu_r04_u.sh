#!/bin/bash
# round 4, call u: the fused C = 64 attention backward again (the kernel lost its read-back, its slow fold and a third of its score-loop instructions since it was last measured)
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3 4; do
  for v in "32" "32,64"; do
    POSE_FUSED_ATTN=$v timeout -k 10 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r04u.json 2> gpurun_out/r04u.err || tail -3 gpurun_out/r04u.err
    echo -n "POSE_FUSED_ATTN=$v "; python scripts/bench_ms.py gpurun_out/r04u.json
  done
done
