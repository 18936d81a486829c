for wgs in 64 128 256; do echo "== C=64 WGS $wgs"; PK_MLP_WGS=$wgs python scripts/bench_kernels.py "block C=64" 2>&1 | grep "bwd_dw"; done
python scripts/bench_kernels.py "block C=64" 2>&1 | grep "FUSED"
python scripts/bench_kernels.py "block C=32" 2>&1 | grep "FUSED"
bash scripts/gpu_round.sh mlp2
