#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag> [1|2|c]
# kernel-trace statistics of the default bench command, then PMC passes (each its own run);
# part 1 = Cornell + Sponza-class, part 2 = the 10 M triangle scene (default: both); c = Cornell alone; m = the Bistro-class frame
# with measured BRDFs (wavefront kernels; one frame per pass)
export TMPDIR=/tmp
TAG=$1
PART=${2:-12}
if [[ $PART == *c* ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_${TAG}_cornell -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/stats_${TAG}_cornell.log 2>&1 || exit 1
bash tools/pmc_sq.sh ${TAG}c || exit 1
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${TAG}c_$P -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/pmc_${TAG}c_$P.log 2>&1 || exit 1
done
echo done cornell
fi
if [[ $PART == *1* ]]; then
S="--workload sponza_like_1920x1080_256spp_envmap_is --steps 2 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_${TAG}_cornell -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/stats_${TAG}_cornell.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_${TAG}_sponza -o stats --output-format csv -- python3 bench.py $S --no-cpu-baseline > gpurun_out/stats_${TAG}_sponza.log 2>&1 || exit 1
bash tools/pmc_sq.sh ${TAG}c || exit 1
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${TAG}c_$P -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/pmc_${TAG}c_$P.log 2>&1 || exit 1
done
bash tools/pmc_mem.sh ${TAG}s $S || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_${TAG}s_WRITE_SIZE -o pmc --output-format csv -- python3 bench.py $S --no-cpu-baseline > gpurun_out/pmc_${TAG}s_WRITE_SIZE.log 2>&1 || exit 1
echo done cornell+sponza
fi
if [[ $PART == *2* ]]; then
Y="--workload courtyard_like_10M_1920x1080_121spp --steps 1 --warmup 1 --no-secondary"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_${TAG}_courtyard -o stats --output-format csv -- python3 bench.py $Y --no-cpu-baseline > gpurun_out/stats_${TAG}_courtyard.log 2>&1 || exit 1
for P in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy VALUUtilization" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 500 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${TAG}y_$N -o pmc --output-format csv -- python3 bench.py $Y --no-cpu-baseline > gpurun_out/pmc_${TAG}y_$N.log 2>&1 || exit 1
done
echo done courtyard
fi
if [[ $PART == *m* ]]; then
M="--workload measured_like_3840x2160_529spp_rgl --steps 1 --warmup 0 --no-secondary --count-sqrt 6"
timeout -k 10 700 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_${TAG}_measured -o stats --output-format csv -- python3 bench.py $M --no-cpu-baseline > gpurun_out/stats_${TAG}_measured.log 2>&1 || exit 1
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "VALUBusy VALUUtilization"; do
  N=$(echo $P | tr " " "_" | cut -c1-40)
  timeout -k 10 700 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${TAG}m_$N -o pmc --output-format csv -- python3 bench.py $M --no-cpu-baseline > gpurun_out/pmc_${TAG}m_$N.log 2>&1 || exit 1
  echo pass $N done
done
echo done measured
fi
