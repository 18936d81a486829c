#!/bin/bash
# round 4, call q: the three bench lines once more (release library, current profiles/), input-pipeline tests, cropper phases
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python scripts/probes/cropper_phases.py 2>&1 | grep -v amdgpu.ids | tail -7
timeout -k 10 300 python -m pytest tests/test_input_pipeline.py -m gpu -q -x --timeout 200 2>&1 | tail -2
timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 30 --warmup 8 > gpurun_out/r04_bench_line_hrnet_w32_384.json 2> gpurun_out/r04_bench_w32.err || tail -3 gpurun_out/r04_bench_w32.err
timeout -k 10 300 python bench.py --config hrformer_base_infer --steps 30 --warmup 5 > gpurun_out/r04_bench_line_hrformer_base_infer.json 2> gpurun_out/r04_bench_base.err || tail -3 gpurun_out/r04_bench_base.err
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err || tail -3 gpurun_out/r04_bench.err
grep "affine_crop\|cpu baseline\|timed region" gpurun_out/r04_bench.err | cut -c1-220
for f in gpurun_out/r04_bench_line*.json; do echo $f; cut -c1-250 $f; echo; done
