#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "${1:-fused or block or attention}" 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>gpurun_out/q3.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
grep "roofline: k_attn_bwd\|roofline: k_mlp" gpurun_out/q3.err | cut -c1-200
done
