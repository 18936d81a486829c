export POSE_FUSED_ATTN=32 POSE_FUSED_ATTN_EVAL=32,64
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=crit
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
python scripts/trace_summary.py $(ls gpurun_out/prof_$tag/*/*kernel_trace.csv | head -1) 3 > gpurun_out/trace_summary_$tag.txt 2>&1
rm -rf gpurun_out/prof_$tag
tail -30 gpurun_out/trace_summary_$tag.txt
