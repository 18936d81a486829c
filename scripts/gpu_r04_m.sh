#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04m_bench.json 2> gpurun_out/r04m_bench.err; echo "bench rc=$?"
grep -E "timed region|roofline: k_conv8p|cpu baseline|failed" gpurun_out/r04m_bench.err | cut -c1-200
python scripts/bench_ms.py gpurun_out/r04m_bench.json
