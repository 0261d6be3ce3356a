#!/bin/bash
# Everything profiles/r4 holds, with ONE build of the library, in one gpurun call.   usage: bash tools/collect_r4.sh   (on the GPU box)
o=gpurun_out/r4/final; mkdir -p $o
step() { echo "== $1"; }
step bench;        python bench.py --steps 100 --warmup 10 > $o/bench_default_run.json 2> $o/bench_default_run.err; tail -c 300 $o/bench_default_run.json; echo
step bench_c4;     python bench.py --backbone resnet101 --planes 10k --steps 40 --no-host-fed --no-cpu-baseline > $o/bench_c4_f16x3.json 2> /dev/null
step bench_c5;     python bench.py --backbone resnet152 --planes 22k --batch 4 --steps 40 --no-host-fed --no-cpu-baseline > $o/bench_c5_f16x3.json 2> /dev/null
step all_dtypes;   python bench.py --steps 30 --all-dtypes --no-cpu-baseline --no-host-fed > $o/bench_all_dtypes.json 2> /dev/null
step profile;      bash tools/profile_bench.sh $o/prof f16x3 > $o/profile_bench.log 2>&1; tail -4 $o/profile_bench.log
step timeline;     bash tools/profile_default_plan.sh $o/prof_default f16x3 > /dev/null 2>&1; tail -1 $o/prof_default/timeline.txt
step pmc;          bash tools/pmc_bench.sh $o/pmc f16x3 > $o/pmc_bench.log 2>&1; tail -12 $o/pmc_bench.log
step floors;       python tools/fill_floor_table.py f16x3 2>/dev/null > $o/small_layer_floors_f16x3.txt; tail -1 $o/small_layer_floors_f16x3.txt
step hbm;          python tools/hbm_layers.py f16x3 2>/dev/null > $o/hbm_layers_f16x3.txt
step x3time;       python tools/bench_conv.py 8 x3time f16x3 3 2>/dev/null > $o/x3time_mix.txt
step pmc_layers;   bash tools/pmc_layers.sh $o/pmc_layers f16x3 > $o/pmc_layers.txt 2>&1; tail -14 $o/pmc_layers.txt
step mfma_valu;    hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o $o/mfma_power 2>/dev/null && (cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $o/mfma_valu -- $o/mfma_power > $o/mfma_valu.log 2>&1)
python3 - $o <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/mfma_valu/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(sys.argv[1] + '/valu_counter_includes_mfma.txt', 'w') as out:
    out.write('register-only MFMA loops of tools/micro/mfma_power.hip under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA:\n')
    for k, v in agg.items():
        va, mf = sum(v['SQ_INSTS_VALU']) / len(v['SQ_INSTS_VALU']), sum(v['SQ_INSTS_MFMA']) / len(v['SQ_INSTS_MFMA'])
        out.write('%-62s SQ_INSTS_VALU %.4g  SQ_INSTS_MFMA %.4g  ratio %.3f\n' % (k, va, mf, va / max(mf, 1)))
print(open(sys.argv[1] + '/valu_counter_includes_mfma.txt').read())
PY
rm -f $o/mfma_power
step corner;       for c in resnet50_1k resnet101_10k resnet152_22k; do python tools/corner_deviation.py --config $c --json $o/corner_deviation_$c.json 2>/dev/null > $o/corner_deviation_$c.txt; done; head -3 $o/corner_deviation_resnet50_1k.txt
python tools/isa_audit.py --json $o/kernel_resources.json | tail -1
