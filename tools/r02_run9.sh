#!/bin/bash
set -o pipefail
O=gpurun_out/r02j
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
for wl in cornell_1024x1024_1024spp_ggx_glass sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp measured_like_3840x2160_529spp_rgl; do
  timeout -k 10 600 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${wl}.json 2> $O/bench_${wl}.err
  echo "$wl rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_${wl}.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
done
