#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant kernel inside bench.py
#   usage (on the GPU box): bash tools/pmc_bench.sh [outdir under gpurun_out] [dtype = f16x3 | bf16 | f16 | bf16x3]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_bench}
dt=${2:-f16x3}
mkdir -p $out
# tile choices from a first un-profiled run, so that the counter passes contain no autotuning launches
export GPP_TUNE_CACHE=$GRAFT_REPO_ROOT/$out/tune_cache.json
python3 bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/tuning_run.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$name -- python3 bench.py --dtype $dt --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-host-fed --no-b1 --repeats 0 > $out/$name.log 2>&1
done
python3 tools/pmc_aggregate.py $out $out/dominant_kernel_pmc_$dt $dt
