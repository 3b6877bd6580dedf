#!/bin/bash
# the two light pdfs of a scatter evaluated in one loop over the lights
set -o pipefail
O=gpurun_out/r02aq
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -1 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for L in lib lib_o lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$L.json 2> $O/bench_co_$L.err || exit 1
  echo "cornell $L: $(python -c "import json; d=json.load(open('$O/bench_co_$L.json')); print(round(d['value'],1))")"
done
for L in lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --workload cornell_256x256_64spp_lambertian --samples-sqrt 32 --no-cpu-baseline > $O/bench_c1_$L.json 2> $O/bench_c1_$L.err || exit 1
  echo "config 1 at 1024 spp $L: $(python -c "import json; d=json.load(open('$O/bench_c1_$L.json')); print(round(d['value'],1))")"
done
