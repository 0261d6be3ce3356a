#!/bin/bash
# Run a list of GPU steps one after the other on the gpurun box: each under its own `timeout -k 10`, output to
# gpurun_out/<name>.log.  An ordinary failure (tests red) does not stop the list; a step that had to be KILLED does
# (never start another GPU step after a hang).   usage: tools/gpu_steps.sh name1 secs1 "cmd1" name2 secs2 "cmd2" ...
mkdir -p gpurun_out
while [ $# -ge 3 ]; do
    name=$1; secs=$2; cmd=$3; shift 3
    echo "== $name: $cmd" | tee gpurun_out/$name.log
    timeout -k 10 $secs bash -c "$cmd" >> gpurun_out/$name.log 2>&1
    rc=$?
    echo "== $name rc=$rc" | tee -a gpurun_out/$name.log
    tail -n 4 gpurun_out/$name.log | cut -c1-400
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed at its limit: stopping"; exit $rc; fi
done
exit 0
