#!/bin/bash
set -o pipefail
O=gpurun_out/r02g
mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "variants or scheduler" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -2 $O/pytest.log
for var in 0 16 8; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --variant $var > $O/bench_cornell_v$var.json 2> $O/bench_cornell_v$var.err
  echo "cornell variant $var rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_cornell_v$var.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
done
