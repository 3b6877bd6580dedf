#!/bin/bash
set -o pipefail
O=gpurun_out/r02m
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo pytest rc $?; tail -2 $O/pytest.log
for wl in cornell_1024x1024_1024spp_ggx_glass sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp; do
  timeout -k 10 600 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  echo "$wl $(python -c "import json; d=json.load(open('$O/bench_$wl.json')); print(round(d['value'],1))")"
done
timeout -k 10 600 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 2 --warmup 1 --no-cpu-baseline --variant 64 > $O/bench_cy_nowalk.json 2>/dev/null
echo "courtyard nowalk $(python -c "import json; d=json.load(open('$O/bench_cy_nowalk.json')); print(round(d['value'],1))")"
timeout -k 10 600 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --steps 2 --warmup 1 --no-cpu-baseline --variant 4 > $O/bench_sp_walk.json 2>/dev/null
echo "sponza walk $(python -c "import json; d=json.load(open('$O/bench_sp_walk.json')); print(round(d['value'],1))")"
