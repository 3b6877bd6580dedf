#!/bin/bash
# experiment: LDS kernel requests both possible next nodes before the box test (second library in lib_b)
set -o pipefail
O=gpurun_out/r02al
mkdir -p $O
WPT_LIB_DIR=$PWD/wurblpt_amd/lib_b timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell_frame or variants or pixel_pool or storage_order or furnace" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -1 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for L in lib lib_b lib lib_b; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$L.json 2> $O/bench_co_$L.err || exit 1
  echo "cornell $L: $(python -c "import json; d=json.load(open('$O/bench_co_$L.json')); print(round(d['value'],1))")"
done
for L in lib lib_b; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 150 python tools/share_cost.py 8 > $O/share8_$L.txt 2>&1 || exit 1
  echo "$L: $(grep 'rank 0' $O/share8_$L.txt)"
done
