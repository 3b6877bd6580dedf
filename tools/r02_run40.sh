#!/bin/bash
# environment map: a sampled bin taken apart by mask and shift where N is a power of two
set -o pipefail
O=gpurun_out/r02aw
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sponza or measured or config_3" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -1 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for L in lib lib_o lib lib_o; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$L.json 2> $O/bench_sp_$L.err || exit 1
  echo "sponza $L: $(python -c "import json; d=json.load(open('$O/bench_sp_$L.json')); print(round(d['value'],1))")"
done
