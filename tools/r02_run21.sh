#!/bin/bash
# experiment: two passes, the second longest pixels first (variant bit 0x40)
set -o pipefail
O=gpurun_out/r02ab
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pixel_pool or variants" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for V in 0 64; do
  timeout -k 10 300 python bench.py --variant $V --no-cpu-baseline > $O/bench_co_$V.json 2> $O/bench_co_$V.err || exit 1
  echo "cornell variant $V: $(python -c "import json; d=json.load(open('$O/bench_co_$V.json')); print(round(d['value'],1))")"
done
for V in 0 64; do
  timeout -k 10 300 python bench.py --variant $V --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$V.json 2> $O/bench_sp_$V.err || exit 1
  echo "sponza variant $V: $(python -c "import json; d=json.load(open('$O/bench_sp_$V.json')); print(round(d['value'],1))")"
done
for V in 0 64; do
  timeout -k 10 600 python bench.py --variant $V --workload courtyard_like_10M_1920x1080_121spp --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cy_$V.json 2> $O/bench_cy_$V.err || exit 1
  echo "courtyard variant $V: $(python -c "import json; d=json.load(open('$O/bench_cy_$V.json')); print(round(d['value'],1))")"
done
