#!/bin/bash
# node steps per look (LDS kernel) and the standing-back threshold, again, on the tree with the pixel pool
set -o pipefail
O=gpurun_out/r02af
mkdir -p $O
for L in lib lib_s2 lib_s4; do
  WPT_LIB_DIR=$PWD/wurblpt_amd/$L timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co_$L.json 2> $O/bench_co_$L.err || exit 1
  echo "cornell $L: $(python -c "import json; d=json.load(open('$O/bench_co_$L.json')); print(round(d['value'],1))")"
done
for V in 4 8 12; do
  timeout -k 10 300 python bench.py --variant $V --no-cpu-baseline > $O/bench_co_v$V.json 2> $O/bench_co_v$V.err || exit 1
  echo "cornell variant $V: $(python -c "import json; d=json.load(open('$O/bench_co_v$V.json')); print(round(d['value'],1))")"
done
