#!/bin/bash
# Kernel trace + stats of bench.py (1 GPU), then the per-launch table of one step and the dominant-kernel summary.
# Usage (on the GPU box):  bash tools/profile_bench.sh [outdir under gpurun_out] [dtype = f16x3 (bench.py's default) | bf16 | f16 | bf16x3]
# A first un-profiled run writes the per-layer tile choices to a cache, so that the profiled run contains no
# autotuning launches and its --stats averages are those of the steady state.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/prof}
dt=${2:-f16x3}
case $dt in bf16) code=1; gf=431.8;; f16) code=2; gf=431.8;; bf16x3) code=4; gf=431.8;; *) code=5; gf=431.8;; esac
xin=""; if [ $code -ge 4 ]; then xin=", true"; fi
fused="0,1"; if [ $code -eq 3 ]; then fused=""; fi; if [ $code -ge 4 ]; then fused="0"; fi      # x3 types: the res2 tails are fused, res3 is not
blocks=""; if [ $code -ge 4 ]; then blocks="0,1"; fi                                               # x3 types: the identity blocks of res2 / res3 are one launch each (GPP_FUSE_BLOCK)
mkdir -p $out
# the per-layer table names the convolution launches by their order in the trace: profiled on ONE stream (the default plan runs res3-res5
# as two half batches on two streams and P5 / P6 / P7 / P4 on side streams, ~4 % faster on the step; the dominant kernel is not touched by that)
export GPP_FPN_LANES=0 GPP_BR1_LANE=0 GPP_HALF_LANES=
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json
python3 bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/tuning_run.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/bench_under_rocprof.log 2>&1
grep '^{"metric"' $out/bench_under_rocprof.log > $out/bench_under_rocprof.json
trace=$(find $out/trace -name '*kernel_trace.csv' | head -1)
stats=$(find $out/trace -name '*kernel_stats.csv' | head -1)
cp "$stats" $out/kernel_stats.csv
python3 tools/trace_table.py "$trace" resnet50 "$fused" "$blocks" > $out/per_layer.txt
tail -3 $out/per_layer.txt
python3 - "$trace" "$code" "$dt" "$xin" > $out/dominant_kernel_trace_summary.txt <<'PY'
import csv, sys
code, dt, xin = sys.argv[2], sys.argv[3], sys.argv[4]
rows = [r for r in csv.DictReader(open(sys.argv[1]))
        if ('conv_igemm_kernel<%s, 256, 256, 2, 4, 2, true%s>' % (code, xin) in r['Kernel_Name'] and int(r['Grid_Size_X']) == 722 * 512) or
           ('conv_igemm_mix_kernel<%s, 256, 224' % code in r['Kernel_Name'] and int(r['Grid_Size_X']) == 748 * 512)]      # (round 4: the mixed-height grid of the same layer)
n_mix = sum('mix_kernel' in r['Kernel_Name'] for r in rows)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d_all = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
d = d_all[9:39]        # bench.py --steps 10 --warmup 3: launches 10..39 are the 30 of the timed region (the HIP events cover those of steps 0, 3, 6, 9)
print('conv_igemm_kernel<%s,256,256,2,4,2,pipe> (grid 722 x 512 threads) / conv_igemm_mix_kernel<%s,256,224> (512 + 236 workgroups) = the regression-tower layers; %d of %d launches in the mixed form' % (dt, dt, n_mix, len(rows)))
print('(3x3, 512->512, five pyramid levels, M = 91504, 431.8 GFLOP per launch), from the rocprofv3 kernel trace:')
print('timed region (30 launches of the 10 timed steps): mean %.1f us  min %.1f us  max %.1f us  ->  %.1f TFLOP/s at the mean' % (sum(d) / len(d), min(d), max(d), 431.8e3 / (sum(d) / len(d))))
print('all %d launches of the run (tuning-free: warm-up + timed steps): mean %.1f us' % (len(d_all), sum(d_all) / len(d_all)))
PY
cat $out/dominant_kernel_trace_summary.txt
