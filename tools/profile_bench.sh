#!/bin/bash
# Kernel trace + stats of bench.py (1 GPU), then the per-launch table of one step.  Usage (on the GPU box):
#   bash tools/profile_bench.sh [outdir under gpurun_out]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/prof}
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $out/bench_under_rocprof.log > $out/bench_under_rocprof.json
trace=$(find $out/trace -name '*kernel_trace.csv' | head -1)
stats=$(find $out/trace -name '*kernel_stats.csv' | head -1)
cp "$stats" $out/kernel_stats.csv
python3 tools/trace_table.py "$trace" > $out/per_layer.txt
tail -4 $out/per_layer.txt
