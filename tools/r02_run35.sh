#!/bin/bash
# the triangle's instance / material / flag words read where the walk found its corners (LDS for small scenes)
set -o pipefail
O=gpurun_out/r02ar
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell_frame or variants or furnace or sphere or texture_probe or mis_test" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -1 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for L in lib lib_o lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$L.json 2> $O/bench_co_$L.err || exit 1
  echo "cornell $L: $(python -c "import json; d=json.load(open('$O/bench_co_$L.json')); print(round(d['value'],1))")"
done
