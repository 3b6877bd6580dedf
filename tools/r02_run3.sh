#!/bin/bash
# round 2: four against five waves per SIMD
set -o pipefail
O=gpurun_out/r02c
mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "variants or scheduler or storage_order or full_size_config_2" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
for var in 0 8; do
  for wl in cornell_1024x1024_1024spp_ggx_glass sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp; do
    timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --variant $var > $O/bench_v${var}_${wl}.json 2> $O/bench_v${var}_${wl}.err
    echo "variant $var $wl rc $?"; python -c "import json,sys; d=json.load(open('$O/bench_v${var}_${wl}.json')); print(d['value'], d['ms_per_step'])"
  done
done
