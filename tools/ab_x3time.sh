#!/bin/bash
# A/B of library builds on the big x3 layers (tools/bench_conv.py x3time):  bash tools/ab_x3time.sh out.txt base variant1 variant2 ...
# ("base" = libgpp_hip.so, anything else = libgpp_hip_<name>.so from `make variant NAME=<name> EXTRA=...`)
out=$1; shift
L=$GRAFT_REPO_ROOT/ground-plane-polling_amd/lib
mkdir -p $(dirname $out)
for v in "$@"; do
  lib=$L/libgpp_hip_$v.so; if [ $v = base ]; then lib=$L/libgpp_hip.so; fi
  echo "== $v"
  GPP_LIB=$lib timeout 200 python3 tools/bench_conv.py 8 x3time ${DT:-f16x3} 3 2>&1 | grep -v amdgpu
done > $out 2>&1
python3 - $out <<'PY'
import sys, re
cur, rows, order = None, {}, []
for line in open(sys.argv[1]):
    if line.startswith('== '):
        cur = line[3:].strip()
        while cur in rows: cur += "'"
        rows[cur] = {}; order.append(cur)
    m = re.match(r'(.+?)\s+\S+ tile\s+(\d+): median\s+([\d.]+) us', line)
    if m: rows[cur][m.group(1).strip()] = float(m.group(3))
    m = re.match(r'sum of medians ([\d.]+)', line)
    if m: rows[cur]['SUM'] = float(m.group(1))
names = list(rows[order[0]].keys())
print('%-24s' % '' + ''.join('%11s' % o[:10] for o in order))
for n in names:
    print('%-24s' % n + ''.join('%11.1f' % rows[o].get(n, float('nan')) for o in order))
PY
