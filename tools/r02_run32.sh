#!/bin/bash
set -o pipefail
O=gpurun_out/r02ao
mkdir -p $O
for L in lib lib_o lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$L.json 2> $O/bench_co_$L.err || exit 1
  echo "cornell $L: $(python -c "import json; d=json.load(open('$O/bench_co_$L.json')); print(round(d['value'],1))")"
done
for L in lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$L.json 2> $O/bench_sp_$L.err || exit 1
  echo "sponza $L: $(python -c "import json; d=json.load(open('$O/bench_sp_$L.json')); print(round(d['value'],1))")"
done
