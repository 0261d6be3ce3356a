#!/bin/bash
# Kernel trace of the DEFAULT plan (side streams, half batches) + its timeline.  usage: bash tools/profile_default_plan.sh outdir [dtype]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/prof_default}; dt=${2:-f16x3}
mkdir -p $out
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json
python3 bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/tuning_run.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/bench_under_rocprof.log 2>&1
trace=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python3 tools/plan_timeline.py "$trace" > $out/timeline.txt
tail -1 $out/timeline.txt
