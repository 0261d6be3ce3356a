#!/bin/bash
# PMC counters of single layers of the f16x3 plan in isolation (review item: FETCH_SIZE, WRITE_SIZE, TCC hit / miss, wait cycles for the layers
# furthest from their floor).   usage (GPU box): bash tools/pmc_layers.sh <outdir> [dtype] [op names ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_layers}; dt=${2:-f16x3}; shift 2
names="${@:-res2b_branch2a res3b_branch2a res3b_branch2c res4b_branch2c C3_reduced pyramid_regression_dim_1}"
mkdir -p $out
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json GPP_HALF_LANES=
python3 tools/run_plan_ops.py $dt $names > $out/times.txt 2>/dev/null
cat $out/times.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$name -- python3 tools/run_plan_ops.py $dt $names > $out/$name.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
# the named ops are the LAST launches of the process: 1 + 8 + 4 = 13 launches per op, in the order given
per_pass = {}
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if any(k in r['Kernel_Name'] for k in ('conv_igemm', 'bottleneck_tail', 'conv1x1_ws'))]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
        by_disp[r['Dispatch_Id']]['_k'] = r['Kernel_Name']; by_disp[r['Dispatch_Id']]['_g'] = r['Grid_Size']
    per_pass[f] = list(by_disp.values())
names = [l.split()[0] for l in open(out + '/times.txt') if 'tile' in l]
n = len(names)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f, disp in per_pass.items():
    tail = disp[-13 * n:]
    for j, name in enumerate(names):
        for k, d in enumerate(tail[13 * j: 13 * (j + 1)]):
            which = 'hot' if 1 <= k <= 8 else ('cold' if k >= 9 else None)
            if which is None:
                continue
            for c, v in d.items():
                if not c.startswith('_'):
                    agg[(name, which)][c].append(v)
            m_ = re.search(r'(conv_igemm\w*<[^>]*>|bottleneck_tail\w*<[^>]*>|conv1x1_ws_kernel<[^>]*>)', d['_k'])
            agg[(name, which)]['_kernel'] = (m_.group(1) if m_ else d['_k'][:60]) + ' grid ' + d['_g']
for (name, which) in sorted(agg):
    a = agg[(name, which)]
    m = {c: sum(v) / len(v) for c, v in a.items() if not c.startswith('_')}
    line = '%-26s %-4s' % (name, which)
    if 'FETCH_SIZE' in m:
        line += '  read %.1f MB (2 x FETCH_SIZE KiB)' % (2 * m['FETCH_SIZE'] * 1024 / 1e6)
    if 'WRITE_SIZE' in m:
        line += '  write %.1f MB' % (m['WRITE_SIZE'] * 1024 / 1e6)
    if 'TCC_HIT_sum' in m:
        line += '  L2 hit %.3f' % (m['TCC_HIT_sum'] / max(1.0, m['TCC_HIT_sum'] + m['TCC_MISS_sum']))
    if 'SQ_WAIT_ANY' in m and 'SQ_WAVE_CYCLES' in m:
        line += '  waves waiting %.2f of their cycles, issuing %.2f' % (m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES'])
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        line += '  MFMA pipe busy %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024))
    if 'SQ_INSTS_VALU' in m and 'SQ_INSTS_MFMA' in m:
        line += '  non-MFMA VALU per MFMA %.2f' % (m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA'] - 1.0)
    print(line + '   [' + a['_kernel'] + ']')
PY
