#!/bin/bash
# usage: tools/pmc_mem.sh <suffix> <bench args...>  -- memory-side counters, separate passes
export TMPDIR=/tmp
SUF=$1; shift
B="python3 bench.py $@ --no-cpu-baseline --no-secondary"
for P in "VALUBusy VALUUtilization" "MeanOccupancyPerCU" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "MemUnitStalled" "TA_BUSY_avr"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${SUF}_$N -o pmc --output-format csv -- $B > gpurun_out/pmc_${SUF}_$N.log 2>&1 || echo "FAILED $P"
done
