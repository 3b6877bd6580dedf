#!/bin/bash
# round 2: scheduler statistics and PMC passes of the rewritten kernel
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 200 python tools/sched_stats.py > $O/sched_cornell.txt 2>&1; echo "sched cornell rc $?"
timeout -k 10 300 python tools/sched_stats.py 0 sponza > $O/sched_sponza.txt 2>&1; echo "sched sponza rc $?"
B="python3 bench.py --no-cpu-baseline --steps 2 --warmup 1"
for P in "VALUBusy VALUUtilization" "MeanOccupancyPerCU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" "WRITE_SIZE" "FETCH_SIZE"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_c_$N -o pmc --output-format csv -- $B > $O/pmc_c_$N.log 2>&1 || echo "FAILED $P"
done
S="--workload sponza_like_1920x1080_256spp_envmap_is"
for P in "VALUBusy VALUUtilization" "WRITE_SIZE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_s_$N -o pmc --output-format csv -- $B $S > $O/pmc_s_$N.log 2>&1 || echo "FAILED $P"
done
ls $O
