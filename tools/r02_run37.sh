#!/bin/bash
set -o pipefail
O=gpurun_out/r02at
mkdir -p $O
timeout -k 10 400 python tools/sched_sweep.py cornell 16 > $O/sweep_cornell.txt 2>&1 || exit 1
tail -1 $O/sweep_cornell.txt; grep default $O/sweep_cornell.txt
for V in 0 4 8 12; do
  timeout -k 10 300 python bench.py --variant $V --no-cpu-baseline > $O/bench_co_v$V.json 2> $O/bench_co_v$V.err || exit 1
  echo "cornell variant $V: $(python -c "import json; d=json.load(open('$O/bench_co_v$V.json')); print(round(d['value'],1))")"
done
