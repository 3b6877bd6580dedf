#!/bin/bash
export TMPDIR=/tmp
SUF=$1; shift
for P in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${SUF}_$N -o pmc --output-format csv -- python3 "$@" > gpurun_out/pmc_${SUF}_$N.log 2>&1 || echo "FAILED $P"
done
