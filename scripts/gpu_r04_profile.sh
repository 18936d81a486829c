#!/bin/bash
# Round-4 evidence: rocprof kernel trace + stats of the default bench command, trace summaries, PMC traffic passes, bench lines.
# Usage (through gpurun, from the repo root): bash scripts/gpu_r04_profile.sh
set -o pipefail
mkdir -p gpurun_out
# heartbeat: long CPU-oracle tests write nothing for minutes; gpurun kills a run that is silent for 7 minutes
( while true; do sleep 60; echo "[heartbeat $(date +%H:%M:%S)]"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof4
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof4.log 2>&1 || { tail -5 gpurun_out/prof4.log; exit 1; }
st=$(find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/prof4 -name "*kernel_trace.csv" | head -1)
cp "$st" gpurun_out/r04_bench_kernel_stats.csv
python scripts/trace_summary.py "$tr" 3 gpurun_out/r04_kernel_union.json > gpurun_out/r04_trace_summary.txt 2>&1 || true
python scripts/profile_meta.py write gpurun_out/r04_profile_meta.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline" > /dev/null
head -12 gpurun_out/r04_trace_summary.txt
rm -rf gpurun_out/prof4/*/*kernel_trace.csv
# PMC traffic: one counter per pass, the program itself after `--`
for mode in isolated step; do
  rm -rf gpurun_out/pmc4_fetch gpurun_out/pmc4_write
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc4_fetch -- python3 scripts/prof_kernels.py $mode > gpurun_out/pmc4_fetch.log 2>&1 || { tail -5 gpurun_out/pmc4_fetch.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc4_write -- python3 scripts/prof_kernels.py $mode > gpurun_out/pmc4_write.log 2>&1 || { tail -5 gpurun_out/pmc4_write.log; exit 1; }
  python scripts/pmc_summary.py $(find gpurun_out/pmc4_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc4_write -name "*counter_collection.csv" | head -1) > gpurun_out/r04_pmc_traffic_$mode.json
  rm -rf gpurun_out/pmc4_fetch gpurun_out/pmc4_write
done
python - <<'PY'
import json
iso = json.load(open("gpurun_out/r04_pmc_traffic_isolated.json"))
stp = json.load(open("gpurun_out/r04_pmc_traffic_step.json"))
k = dict(stp["kernels"])
k.update(iso["kernels"])           # the isolated pass wins for the kernels it covers (one shape per kernel name)
json.dump({"note": iso["note"], "sources": "isolated probes (scripts/prof_kernels.py isolated) over a whole eager step (… step)", "kernels": k},
          open("gpurun_out/r04_pmc_traffic.json", "w"), indent=1)
for n in ("k_conv8p", "k_wgrad3", "k_conv3h<64, 64>", "k_igemm2<128, 32, 4, 1, 64, 0>", "k_reduce_many", "k_attn_bwd<32>", "k_mlp_fwd<32>"):
    print(n, k.get(n))
PY
sed "s/r03_pmc_sq_counters/r04_pmc_sq_counters/g" scripts/gpu_pmc_sq.sh > gpurun_out/_sq.sh; bash gpurun_out/_sq.sh > gpurun_out/r04_sq.log 2>&1 || tail -3 gpurun_out/r04_sq.log
sed "s/r03_pmc_traffic_wide/r04_pmc_traffic_wide/g" scripts/gpu_pmc_traffic_wide.sh > gpurun_out/_wide.sh; bash gpurun_out/_wide.sh > gpurun_out/r04_wide.log 2>&1 || tail -3 gpurun_out/r04_wide.log
bash scripts/gpu_sections.sh > gpurun_out/r04_sections.log 2>&1 || tail -3 gpurun_out/r04_sections.log
# bench lines (the default line last: it is the one the roofline entries read profiles/ for)
timeout -k 10 300 python bench.py --config hrnet_w32_384 --steps 30 --warmup 8 > gpurun_out/r04_bench_line_hrnet_w32_384.json 2> gpurun_out/r04_bench_w32.err || tail -3 gpurun_out/r04_bench_w32.err
timeout -k 10 300 python bench.py --config hrformer_base_infer --steps 30 --warmup 5 > gpurun_out/r04_bench_line_hrformer_base_infer.json 2> gpurun_out/r04_bench_base.err || tail -3 gpurun_out/r04_bench_base.err
cp gpurun_out/r04_bench_kernel_stats.csv gpurun_out/r04_kernel_union.json gpurun_out/r04_profile_meta.json gpurun_out/r04_pmc_traffic_isolated.json gpurun_out/r04_pmc_traffic_step.json gpurun_out/r04_pmc_traffic_wide.json profiles/ 2>/dev/null
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err || tail -3 gpurun_out/r04_bench.err
grep "roofline\|cpu baseline\|timed region" gpurun_out/r04_bench.err | cut -c1-220
for f in gpurun_out/r04_bench_line*.json; do echo $f; cut -c1-330 $f; echo; done
