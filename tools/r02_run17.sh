#!/bin/bash
set -o pipefail
O=gpurun_out/r02u
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -1 $O/pytest.log
for V in 0 4; do
  timeout -k 10 300 python bench.py --variant $V --no-cpu-baseline > $O/bench_co_$V.json 2> $O/bench_co_$V.err || exit 1
  echo "cornell variant $V: $(python -c "import json; d=json.load(open('$O/bench_co_$V.json')); print(round(d['value'],1))")"
done
for V in 0 4; do
  timeout -k 10 300 python bench.py --variant $V --workload cornell_256x256_64spp_lambertian --samples-sqrt 32 --no-cpu-baseline > $O/bench_c1_$V.json 2> $O/bench_c1_$V.err || exit 1
  echo "config 1 variant $V: $(python -c "import json; d=json.load(open('$O/bench_c1_$V.json')); print(round(d['value'],1))")"
done
