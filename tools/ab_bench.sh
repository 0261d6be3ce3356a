#!/bin/bash
# Same-box A/B of two builds of the library (boxes of the pool differ by several percent, so numbers from two gpurun calls
# cannot be compared): alternates `bench.py` between GPP_LIB=$1 and GPP_LIB=$2, $3 rounds (default 3), core loop only.
#   usage: tools/ab_bench.sh path/to/libA.so path/to/libB.so [rounds] [extra bench.py args ...]
# AB_ENV_A / AB_ENV_B: optional "NAME=value NAME2=value" settings for the A / B runs (the same library may be given twice).
A=$1; B=$2; R=${3:-3}; shift 3
mkdir -p gpurun_out
for r in $(seq 1 $R); do
    for which in A B; do
        lib=$A; [ $which = B ] && lib=$B
        extra=$AB_ENV_A; [ $which = B ] && extra=$AB_ENV_B
        env $extra GPP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 --steps 40 "$@" > gpurun_out/ab_$which$r.json 2> gpurun_out/ab_$which$r.err || { echo "bench failed ($which$r)"; tail -5 gpurun_out/ab_$which$r.err; exit 1; }
        python - <<P
import json
d = json.load(open('gpurun_out/ab_$which$r.json'))
print('$which$r', '$lib', d['value'], 'images/s', d['ms_per_step'], 'ms/step; dominant launch', d['roofline']['mean_launch_ms'], 'ms')
P
    done
done
