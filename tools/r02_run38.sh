#!/bin/bash
# does the kernel's own copy of the scene view (materials in LDS) cost the kernels that fetch the scene from HBM anything?
set -o pipefail
O=gpurun_out/r02au
mkdir -p $O
for L in lib lib_o lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$L.json 2> $O/bench_sp_$L.err || exit 1
  echo "sponza $L: $(python -c "import json; d=json.load(open('$O/bench_sp_$L.json')); print(round(d['value'],1))")"
done
