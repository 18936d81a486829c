#!/bin/bash
# A/B of env settings on the default bench line: each argument is one "VAR=value[,VAR=value]" set; "-" = defaults
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
for setting in "$@"; do
  envs=""
  if [ "$setting" != "-" ]; then envs=$(echo "$setting" | tr ',' ' '); fi
  r=$(env $envs timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>gpurun_out/envab.err | grep -o '"ms_per_step": [0-9.]*')
  red=$(grep "roofline: k_reduce_many" gpurun_out/envab.err | grep -o "[0-9.]* us isolated" | head -1)
  echo "$setting  $r  reduce_many $red"
done
done
