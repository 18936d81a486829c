#!/bin/bash
# A/B of env settings on the cfg-4 bench line (HRNet-W32 training): each argument "VAR=value[,VAR=value]"; "-" = defaults
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
for setting in "$@"; do
  envs=""
  if [ "$setting" != "-" ]; then envs=$(echo "$setting" | tr ',' ' '); fi
  r=$(env $envs timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>gpurun_out/envab4.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$setting  $r"
done
done
