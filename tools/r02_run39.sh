#!/bin/bash
# the camera rays' common origin evaluated once per workgroup (kernels without lens features)
set -o pipefail
O=gpurun_out/r02av
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -1 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for L in lib lib_o lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$L.json 2> $O/bench_co_$L.err || exit 1
  echo "cornell $L: $(python -c "import json; d=json.load(open('$O/bench_co_$L.json')); print(round(d['value'],1))")"
done
