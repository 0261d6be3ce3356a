#!/bin/bash
# Per-stage split of a step from roctx ranges, no kernel-name matching: with GPP_ROCTX=2 gpp_plan_run opens a range per stage of the plan ("gpp:stem",
# "gpp:backbone", "gpp:fpn", "gpp:heads", "gpp:decode", "gpp:polling") and synchronises the device at the range boundaries, so a range's duration in
# rocprofv3's marker trace is that stage's time on the device.   usage (GPU box): bash tools/roctx_stages.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/roctx}; mkdir -p $out
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/tuning_run.log 2>&1
GPP_ROCTX=2 timeout 600 rocprofv3 --marker-trace --output-format csv -d $out/trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/bench_under_rocprof.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + '/trace/**/*marker_api_trace.csv', recursive=True)
if not files:
    print('no marker trace found under', out); sys.exit(1)
rows = sorted((r for r in csv.DictReader(open(files[0])) if r['Function'].startswith('gpp:')), key=lambda r: int(r['Start_Timestamp']))
# a whole plan run = the ranges from one "gpp:stem" up to the next that also hold "gpp:polling" (the tile tuner's single-op runs of the plan build are the short ones)
runs, cur = [], []
for r in rows:
    if r['Function'] == 'gpp:stem' and cur:
        runs.append(cur); cur = []
    cur.append((r['Function'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
runs.append(cur)
runs = [x for x in runs if any(n == 'gpp:polling' for n, _ in x) and len(x) >= 6]
runs = runs[-8:]                                       # the timed steps (the first whole runs are warm-up)
print('stages of a plan run (GPP_ROCTX=2: the device is synchronised at the range boundaries, so a range lasts as long as its stage does on the device;')
print('the default plan -- side lanes, half batches -- with every lane joined at the boundaries); median over the last %d plan runs of the trace' % len(runs))
names = []
for x in runs:
    for n, _ in x:
        if n not in names:
            names.append(n)
total = 0.0
for n in names:
    per_run = sorted(sum(us for m, us in x if m == n) for x in runs)
    med = per_run[len(per_run) // 2]
    total += med
    print('  %-14s %9.1f us   (min %.1f, max %.1f)' % (n, med, per_run[0], per_run[-1]))
print('  %-14s %9.1f us   (a step of the un-synchronised plan is shorter: the stages overlap at their seams)' % ('sum', total))
PY
