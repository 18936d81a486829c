#!/bin/bash
# quick step-level check: a few parity tests + two bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_network_ops.py -m gpu -q -x --timeout 300 -k "${1:-deferred or trainer_two or wgrad or conv_fwd}" 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>gpurun_out/q2.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done
grep "roofline: k_reduce_many\|roofline: k_conv8p\|roofline: k_attn_bwd" gpurun_out/q2.err | cut -c1-200
