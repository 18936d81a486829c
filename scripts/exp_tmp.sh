bash scripts/gpu_round.sh r2b || exit 1
timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 20 --warmup 5 > gpurun_out/bench_w32.json 2> gpurun_out/bench_w32.err; tail -c 700 gpurun_out/bench_w32.json; tail -3 gpurun_out/bench_w32.err
timeout -k 10 300 python bench.py --config hrformer_base_infer --steps 20 --warmup 5 > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err; tail -c 700 gpurun_out/bench_base.json; tail -3 gpurun_out/bench_base.err
timeout -k 10 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 3000 gpurun_out/bench_full.json; grep roofline gpurun_out/bench_full.err
