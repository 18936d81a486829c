#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for cfg in "512 256 512" "512 256 256" "512 512 512" "1024 256 512" "1024 256 256" "512 256 1024" "256 256 512"; do set -- $cfg
PK_WGRAD4_WGS=$1 PK_WGRAD4_WGS9=$2 PK_WGRAD4_ROWS=$3 timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>gpurun_out/sw.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' || { tail -5 gpurun_out/sw.err; exit 1; }; echo " wgs=$1 wgs9=$2 rows=$3"; done
