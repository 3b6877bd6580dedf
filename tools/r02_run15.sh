#!/bin/bash
# experiment: kinds of material with few lanes in a long round stand back once (variant bits 2-3: fewer than 3 / 6 / 9 lanes)
set -o pipefail
O=gpurun_out/r02s
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variant or scheduler or kernel_choice" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -1 $O/pytest.log
for V in 0 4 8 12; do
  timeout -k 10 300 python bench.py --variant $V --no-cpu-baseline > $O/bench_co_$V.json 2> $O/bench_co_$V.err || exit 1
  echo "cornell wait $V: $(python -c "import json; d=json.load(open('$O/bench_co_$V.json')); print(round(d['value'],1))")"
done
for V in 0 4 8; do
  timeout -k 10 300 python bench.py --variant $V --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$V.json 2> $O/bench_sp_$V.err || exit 1
  echo "sponza wait $V: $(python -c "import json; d=json.load(open('$O/bench_sp_$V.json')); print(round(d['value'],1))")"
done
for V in 4 8 12; do
python tools/sched_stats.py $V > $O/sched_$V.txt 2>&1
grep "SHADE:" $O/sched_$V.txt
done
