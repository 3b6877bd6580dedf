#!/bin/bash
set -o pipefail
O=gpurun_out/r02aa
mkdir -p $O
for G in 4 3 2; do
  WPT_POOL_GROUPS_PER_CU=$G timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$G.json 2> $O/bench_co_$G.err || exit 1
  echo "cornell groups/CU $G: $(python -c "import json; d=json.load(open('$O/bench_co_$G.json')); print(round(d['value'],1))")"
done
for G in 3 2; do
  WPT_POOL_GROUPS_PER_CU=$G timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$G.json 2> $O/bench_sp_$G.err || exit 1
  echo "sponza groups/CU $G: $(python -c "import json; d=json.load(open('$O/bench_sp_$G.json')); print(round(d['value'],1))")"
done
WPT_POOL_GROUPS_PER_CU=3 timeout -k 10 600 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cy_3.json 2> $O/bench_cy_3.err || exit 1
echo "courtyard groups/CU 3: $(python -c "import json; d=json.load(open('$O/bench_cy_3.json')); print(round(d['value'],1))")"
