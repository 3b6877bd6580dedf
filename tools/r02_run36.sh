#!/bin/bash
# material records in LDS (variant bit 0x80: not)
set -o pipefail
O=gpurun_out/r02as
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -1 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for V in 0 128 0 128; do
  timeout -k 10 300 python bench.py --variant $V --no-cpu-baseline > $O/bench_co_$V.json 2> $O/bench_co_$V.err || exit 1
  echo "cornell variant $V: $(python -c "import json; d=json.load(open('$O/bench_co_$V.json')); print(round(d['value'],1))")"
done
