#!/bin/bash
# PMC passes for one conv shape. usage: tools/pmc_conv.sh <outdir> <tile> [cin cout k]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 tools/bench_one.py "$@" > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    v = agg[k]
    print('%-34s n=%d  mean %.4g' % (k, len(v), sum(v) / len(v)))
PY
