#!/bin/bash
# usage: tools/pmc_sq.sh <outdir-suffix> <bench args...>   (runs on the GPU box; separate --pmc passes)
export TMPDIR=/tmp
SUF=$1; shift
B="python3 bench.py $@ --no-cpu-baseline --no-secondary"
for P in "VALUBusy VALUUtilization" "MeanOccupancyPerCU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${SUF}_$N -o pmc --output-format csv -- $B > gpurun_out/pmc_${SUF}_$N.log 2>&1 || echo "FAILED $P"
done
