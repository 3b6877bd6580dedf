#!/bin/bash
# round 2, first GPU call: parity of the rewritten kernel, then first rates (default build vs build without machine LICM)
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for lib in lib lib_nolicm; do
  for wl in cornell_1024x1024_1024spp_ggx_glass sponza_like_1920x1080_256spp_envmap_is; do
    WPT_LIB_DIR=$lib timeout -k 10 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${lib}_${wl}.json 2> $O/bench_${lib}_${wl}.err
    echo "$lib $wl rc $?"; python -c "import json,sys; d=json.load(open('$O/bench_${lib}_${wl}.json')); print(d['value'], d['ms_per_step'])"
  done
done
for top in 0 65536 1048576; do
  WPT_LIB_DIR=lib timeout -k 10 400 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 2 --warmup 1 --no-cpu-baseline --top-nodes $top > $O/bench_courtyard_top$top.json 2> $O/bench_courtyard_top$top.err
  echo "courtyard top $top rc $?"; python -c "import json,sys; d=json.load(open('$O/bench_courtyard_top$top.json')); print(d['value'], d['ms_per_step'])"
done
