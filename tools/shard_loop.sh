#!/bin/bash
# repeat the two-process shard test until it mismatches (debugging aid); leaves the dumps under gpurun_out/
for i in 1 2 3 4; do
  python -m pytest tests/test_polling_gpu.py tests/test_zz_sharded_gpu.py -q -k "sharded or db" -x > gpurun_out/shardloop_$i.log 2>&1
  rc=$?
  tail -n 3 gpurun_out/shardloop_$i.log | cut -c1-800
  if [ $rc -ne 0 ]; then grep -E "AssertionError" gpurun_out/shardloop_$i.log | cut -c1-1500; exit 0; fi
done
