#!/bin/bash
# usage (GPU box): bash tools/pmc_memory_pipeline.sh <suffix> <bench.py arguments...>
# Where a walk's fetches wait: address / data path (TA, TD), the vector L1 (TCP) and the L2 (TCC), one rocprofv3 --pmc pass per
# group of counters (never together with other traces); tools/pmc_sum.py <suffix> prints the sums per kernel.
# Round 3 on this pool: only the two TCP groups of four counters came back (profiles/r03_pmc_memory_pipeline_sponza64spp.txt);
# the groups with GRBM_GUI_ACTIVE / TA_* / TD_* / TCC_* and the TCP group of five counters ended in rocprofv3's abort handler,
# three of them only after their 300 s limit -- run the groups one per gpurun call, or the call's own limit goes on them.
export TMPDIR=/tmp
SUF=$1; shift
for P in \
  "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
  "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
  "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
  "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_READ_SECTORS_sum" \
  "TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_BUBBLE_sum TCC_READ_sum"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${SUF}_$N -o pmc --output-format csv -- python3 bench.py "$@" > gpurun_out/pmc_${SUF}_$N.log 2>&1 || echo "FAILED $P"
  echo "pass $N done"
done
