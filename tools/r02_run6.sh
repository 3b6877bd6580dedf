#!/bin/bash
set -o pipefail
O=gpurun_out/r02f
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
for var in 0 16; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --variant $var > $O/bench_cornell_v$var.json 2> $O/bench_cornell_v$var.err
  echo "cornell variant $var rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_cornell_v$var.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
done
for wl in sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp; do
  timeout -k 10 500 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${wl}.json 2> $O/bench_${wl}.err
  echo "$wl rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_${wl}.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
done
