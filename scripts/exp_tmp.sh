#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/bench_tk.json 2> gpurun_out/bench_tk.err || { tail -5 gpurun_out/bench_tk.err; exit 1; }
grep "roofline\|time_kernel" gpurun_out/bench_tk.err | cut -c1-200
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_tk.json | head -2
