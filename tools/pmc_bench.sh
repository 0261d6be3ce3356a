#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant kernel inside bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_bench}
mkdir -p $out
# tile choices from a first un-profiled run, so that the counter passes contain no autotuning launches
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed > $out/tuning_run.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$name -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-host-fed > $out/$name.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_igemm_kernel<1, 256, 256' in k and int(r['Grid_Size']) == 722 * 512:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('dominant kernel = conv_igemm_kernel<bf16,256,256,2,4,2,pipe>, grid 722 x 512 (regression tower 3x3 512->512, M=91504)')
for k in sorted(agg):
    v = agg[k]
    print('%-30s launches=%d  mean per launch %.6g' % (k, len(v), sum(v) / len(v)))
PY
