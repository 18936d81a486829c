#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py --config hrformer_base_infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_base_r02.json 2> gpurun_out/bench_base_r02.err || { tail -5 gpurun_out/bench_base_r02.err; exit 1; }
cat gpurun_out/bench_base_r02.json | cut -c1-700; grep -i "capture\|graph" gpurun_out/bench_base_r02.err | head -5
timeout -k 10 400 python bench.py --config hrnet_w32_384 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_w32_r02.json 2> gpurun_out/bench_w32_r02.err || { tail -5 gpurun_out/bench_w32_r02.err; exit 1; }
cat gpurun_out/bench_w32_r02.json | cut -c1-700
