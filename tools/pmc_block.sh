#!/bin/bash
# Fabric-side bytes of one bottleneck as three launches against gpp_bottleneck_block (tools/bench_block.py under rocprofv3, one counter group per pass).
#   usage (GPU box): bash tools/pmc_block.sh <outdir> [stage=3] [B=8] [tile=0]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_block}; stage=${2:-3}; B=${3:-8}; tile=${4:-0}
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$name -- python3 tools/bench_block.py $stage $B f16x3 4 $tile > $out/$name.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        k = r['Kernel_Name']
        key = 'block' if 'bottleneck_block' in k else ('ws1x1' if 'conv1x1_ws' in k else ('igemm grid ' + r['Grid_Size'] if 'conv_igemm' in k else None))
        if key:
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in sorted(agg):
    a = agg[key]
    # the last launches of every kind are the timed ones (the autotuner's come first): median is robust to them
    med = {c: sorted(v)[len(v) // 2] for c, v in a.items()}
    line = '%-28s launches %4d' % (key, max(len(v) for v in a.values()))
    if 'FETCH_SIZE' in med: line += '  read %.1f MB (2 x FETCH_SIZE)' % (2 * med['FETCH_SIZE'] * 1024 / 1e6)
    if 'WRITE_SIZE' in med: line += '  write %.1f MB' % (med['WRITE_SIZE'] * 1024 / 1e6)
    if 'TCC_HIT_sum' in med: line += '  L2 hit %.3f' % (med['TCC_HIT_sum'] / max(1.0, med['TCC_HIT_sum'] + med['TCC_MISS_sum']))
    if 'SQ_WAIT_ANY' in med: line += '  waves waiting %.2f, issuing %.2f' % (med['SQ_WAIT_ANY'] / med['SQ_WAVE_CYCLES'], med['SQ_ACTIVE_INST_ANY'] / med['SQ_WAVE_CYCLES'])
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in med: line += '  MFMA busy %.3f' % (med['SQ_VALU_MFMA_BUSY_CYCLES'] / (med['GRBM_GUI_ACTIVE'] / 8 * 1024))
    print(line)
PY
