#!/bin/bash
set -o pipefail
O=gpurun_out/r02r
mkdir -p $O
WPT_NODE_BLOCK_LEVELS=3 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "storage_order or courtyard_like or sponza_like_textures" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -1 $O/pytest.log
for L in 0 2 3 4 6; do
  WPT_NODE_BLOCK_LEVELS=$L timeout -k 10 600 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cy_L$L.json 2> $O/bench_cy_L$L.err
  echo "courtyard block levels $L: $(python -c "import json; d=json.load(open('$O/bench_cy_L$L.json')); print(round(d['value'],1))")"
done
for L in 3 4; do
  WPT_NODE_BLOCK_LEVELS=$L timeout -k 10 600 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_sp_L$L.json 2> $O/bench_sp_L$L.err
  echo "sponza block levels $L: $(python -c "import json; d=json.load(open('$O/bench_sp_L$L.json')); print(round(d['value'],1))")"
done
