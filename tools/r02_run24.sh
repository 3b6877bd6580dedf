#!/bin/bash
set -o pipefail
O=gpurun_out/r02ae
mkdir -p $O
for R in 0 1; do
  if [ $R = 1 ]; then export WPT_POOL_REVERSE=1; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$R.json 2> $O/bench_co_$R.err || exit 1
  echo "cornell reverse $R: $(python -c "import json; d=json.load(open('$O/bench_co_$R.json')); print(round(d['value'],1))")"
  timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$R.json 2> $O/bench_sp_$R.err || exit 1
  echo "sponza reverse $R: $(python -c "import json; d=json.load(open('$O/bench_sp_$R.json')); print(round(d['value'],1))")"
done
