#!/bin/bash
# A/B of two library builds on one box: default libposekernels.so vs csrc/libposekernels_b.so (POSE_KERNELS_LIB), alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
B="$GRAFT_REPO_ROOT/infantposeestimation_gaussianbias_amd/csrc/libposekernels_b.so"
if [ -n "$1" ]; then timeout -k 10 900 python -m pytest tests/test_gpu_network_ops.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "$1" 2>&1 | tail -4 || exit 1; fi
for v in A B A B; do
  if [ $v = B ]; then export POSE_KERNELS_LIB="$B"; else unset POSE_KERNELS_LIB; fi
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/lib=$v  /"; echo
done
