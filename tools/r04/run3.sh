#!/bin/bash
# round 4, third GPU call: the GPU suite, the default bench line with its three secondaries (how long it takes), then what the
# memory side and the vector pipes say about the wide walk against the binary walk (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; one
# rocprofv3 --pmc pass each, program directly behind --)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
T0=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print('primary', round(d['value'],1), d['parity']['bits_differ'], d['roofline']['frac'])
for s in d['secondary']: print(s['workload'], round(s['value'],1), round(s['ms_per_step'],1), s['parity']['bits_differ'], round(s['roofline']['frac'],3), s['roofline']['kernel'], round(s['cpu_baseline']['value'],2), s['roofline']['counted_on'])
"
Y="--workload courtyard_like_10M_1920x1080_121spp --samples-sqrt 6 --steps 1 --warmup 0 --no-cpu-baseline"
S="--workload sponza_like_1920x1080_256spp_envmap_is --samples-sqrt 8 --steps 1 --warmup 0 --no-cpu-baseline"
for W in y s; do
  if [ $W = y ]; then A="$Y"; else A="$S"; fi
  for V in bin wide; do
    if [ $V = wide ]; then X="--wide-walk"; else X=""; fi
    for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
      N=$(echo $P | tr " " "_" | cut -c1-30)
      timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_${W}_${V}_$N -o pmc --output-format csv -- python3 bench.py $A $X > $O/pmc_${W}_${V}_$N.log 2>&1 || { echo "FAILED $W $V $P"; tail -3 $O/pmc_${W}_${V}_$N.log; exit 1; }
    done
    echo "pmc $W $V done"
  done
done
