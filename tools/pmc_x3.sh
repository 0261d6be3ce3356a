#!/bin/bash
# PMC counters of the dominant layer of the bf16x3 path inside bench.py --dtype bf16x3 (where do its wavefronts spend their cycles?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_x3}
mkdir -p $out
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json
python3 bench.py --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/tuning_run.log 2>&1
tail -c 600 $out/tuning_run.log
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$name -- python3 bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/$name.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_igemm_kernel<4' in k and int(r['Grid_Size']) > 100000:
            agg[(k[k.index('<'):k.index('>') + 1], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for key in sorted(agg, key=lambda k: -sum(agg[k].get('SQ_BUSY_CYCLES', [0])))[:4]:
    print(key)
    for c in sorted(agg[key]):
        v = agg[key][c]
        print('   %-30s n=%3d mean %.6g' % (c, len(v), sum(v) / len(v)))
PY
