#!/bin/bash
# Everything profiles/r6 holds that comes from the GPU, with ONE build of the library.   usage (on the GPU box):
#   bash tools/collect_r6.sh a      (a1 + a2) bench lines (default 100 steps, driver style, configs 4 / 5), B = 1 latency + per-layer floors, rocprofv3 trace + per-layer table, PMC passes
#   bash tools/collect_r6.sh b      the GPU suite: default selection (timed against its budget), then the `slow` selection
part=${1:-a}
o=gpurun_out/r6/final; mkdir -p $o
step() { echo "== $1"; }
if [ "$part" = a ] || [ "$part" = a1 ]; then
step bench;        python bench.py --steps 100 --warmup 10 > $o/bench_default_run.json 2> $o/bench_default_run.err; tail -c 300 $o/bench_default_run.json; echo
step driver_style; python bench.py > $o/bench_driver_style.json 2> $o/bench_driver_style.err; tail -c 200 $o/bench_driver_style.json; echo
step bench_c4;     python bench.py --backbone resnet101 --planes 10k --steps 40 --no-host-fed --no-cpu-baseline --no-b1 > $o/bench_c4_f16x3.json 2> /dev/null
step bench_c5;     python bench.py --backbone resnet152 --planes 22k --batch 4 --steps 40 --no-host-fed --no-cpu-baseline --no-b1 > $o/bench_c5_f16x3.json 2> /dev/null
step b1;           python tools/b1_latency.py --n 100 > $o/b1_latency_default_plan.json 2> $o/b1_latency.err; GPP_PLAN=latency python tools/b1_latency.py --n 100 > $o/b1_latency_latency_plan.json 2>> $o/b1_latency.err; tail -c 300 $o/b1_latency_latency_plan.json; echo
step b1_floors;    python tools/fill_floor_table.py f16x3 resnet50 1 2>/dev/null > $o/b1_per_layer.txt; tail -1 $o/b1_per_layer.txt
fi
if [ "$part" = a ] || [ "$part" = a2 ]; then
step profile;      bash tools/profile_bench.sh $o/prof f16x3 > $o/profile_bench.log 2>&1; tail -4 $o/profile_bench.log
step pmc;          bash tools/pmc_bench.sh $o/pmc f16x3 > $o/pmc_bench.log 2>&1; tail -12 $o/pmc_bench.log
step block;        for st in 2 3; do python tools/bench_block.py $st 8 f16x3 30 0,1814 > $o/bench_block_res${st}_b8.txt 2>&1; tail -2 $o/bench_block_res${st}_b8.txt | cut -c1-260; done
step roctx;        bash tools/roctx_stages.sh $o/roctx > $o/roctx_stages.log 2>&1; tail -8 $o/roctx_stages.log
python tools/isa_audit.py --json $o/kernel_resources.json | tail -1
fi
if [ "$part" = b ]; then
step smoke;        python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-300
step gpu_suite;    GPP_ENFORCE_SUITE_BUDGET=1 python -m pytest tests -m gpu -q --durations=15 > $o/gpu_suite_default.log 2>&1; tail -40 $o/gpu_suite_default.log
step slow_suite;   python -m pytest tests -m "gpu and slow" -q --durations=10 > $o/gpu_suite_slow.log 2>&1; tail -14 $o/gpu_suite_slow.log
fi
