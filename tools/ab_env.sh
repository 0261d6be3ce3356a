#!/bin/bash
# Same-box A/B of plan options given as environment settings (one library):  bash tools/ab_env.sh out.txt rounds "NAME=v ..." "NAME=w ..." ...
# Runs bench.py (core loop only, 40 steps) once per setting and round, alternating; "-" = no setting (the default plan).
out=$1; R=$2; shift 2
mkdir -p $(dirname $out)
: > $out
for r in $(seq 1 $R); do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    s="$setting"; [ "$s" = "-" ] && s=""
    env $s python bench.py --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 --steps 40 ${AB_ARGS} > gpurun_out/ab_env_$i.json 2> gpurun_out/ab_env_$i.err || { echo "bench failed ($setting)" >> $out; tail -3 gpurun_out/ab_env_$i.err >> $out; continue; }
    python - "$setting" $r >> $out <<'P'
import json, sys, glob
d = json.load(open(sorted(glob.glob('gpurun_out/ab_env_*.json'), key=lambda p: __import__('os').path.getmtime(p))[-1]))
print('round %s  %-44s %8.2f images/s  %7.3f ms/step  dominant launch %.4f ms' % (sys.argv[2], sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms']))
P
  done
done
cat $out
