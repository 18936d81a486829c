timeout -k 10 300 python -m pytest tests/test_gpu_network_ops.py -m gpu -q -x --timeout 120 -k "conv_fwd_dgrad_wgrad" 2>&1 | tail -3 | cut -c1-300
for w in 256 288 512; do echo "wide wgs $w"; PK_WGRAD_WIDE_WGS=$w python scripts/bench_kernels.py "256->256 k3" 2>&1 | grep wgrad; done
