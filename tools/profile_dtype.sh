#!/bin/bash
# Kernel trace of bench.py at another arithmetic type (f32 | bf16x3 | f16): per-launch table of one step + the JSON line.
# Usage (on the GPU box):  bash tools/profile_dtype.sh f32 [outdir under gpurun_out]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
dt=${1:-f32}
out=${2:-gpurun_out/prof_$dt}
mkdir -p $out
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json
python3 bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/tuning_run.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --dtype $dt --steps 6 --warmup 2 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/bench_under_rocprof.log 2>&1
grep '^{"metric"' $out/bench_under_rocprof.log > $out/bench_under_rocprof.json
trace=$(find $out/trace -name '*kernel_trace.csv' | head -1)
fused="0,1"; if [ "$dt" = "f32" ]; then fused=""; fi; if [ "$dt" = "bf16x3" ] || [ "$dt" = "f16x3" ]; then fused="0"; fi
python3 tools/trace_table.py "$trace" resnet50 "$fused" > $out/per_layer.txt
tail -3 $out/per_layer.txt
rm -rf $out/trace
