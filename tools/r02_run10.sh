#!/bin/bash
set -o pipefail
O=gpurun_out/r02k
mkdir -p $O
for var in 0 64; do
  timeout -k 10 600 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 2 --warmup 1 --no-cpu-baseline --variant $var > $O/bench_courtyard_v$var.json 2> $O/bench_courtyard_v$var.err
  echo "courtyard variant $var rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_courtyard_v$var.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
done
for var in 0 4; do
  timeout -k 10 600 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --steps 2 --warmup 1 --no-cpu-baseline --variant $var > $O/bench_sponza_v$var.json 2> $O/bench_sponza_v$var.err
  echo "sponza variant $var rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_sponza_v$var.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
done
