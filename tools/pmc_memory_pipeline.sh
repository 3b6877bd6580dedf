#!/bin/bash
# usage (GPU box): bash tools/pmc_memory_pipeline.sh <suffix> <bench.py arguments...>
# Where a walk's fetches wait: address / data path (TA, TD), the vector L1 (TCP) and the L2 (TCC), one rocprofv3 --pmc pass per
# group of counters (never together with other traces); tools/pmc_sum.py <suffix> prints the sums per kernel.
# Round 3 asked for five counters of one block in a pass, and rocprofv3 refused them before the first kernel ran
# ("rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect", fatal:
# gpurun_out/pmc_mp_TCC_BUSY_sum_TCC_CYCLE_sum_TCC_TAG_STALL.log of that round): a block has a fixed number of counter slots
# per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots": TCC 4, GRBM 2; the two TCP groups of four came back, the group of five did
# not).  The groups below hold at most four counters of TCP or TCC and at most two of TA, TD or GRBM, one block per pass; the
# list of counters the tool knows on this box goes to gpurun_out/pmc_<suffix>_counters.txt first.  A pass that fails ends the
# script (no further GPU step behind a failed one).
export TMPDIR=/tmp
SUF=$1; shift
rocprofv3 -L > gpurun_out/pmc_${SUF}_counters.txt 2>&1 || true
for P in \
  "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
  "TCC_EA0_RDREQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
  "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
  "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  if ! timeout -k 10 240 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${SUF}_$N -o pmc --output-format csv -- python3 bench.py "$@" > gpurun_out/pmc_${SUF}_$N.log 2>&1; then
    echo "FAILED $P (see gpurun_out/pmc_${SUF}_$N.log); stopping"
    tail -3 gpurun_out/pmc_${SUF}_$N.log
    exit 1
  fi
  echo "pass $N done"
done
