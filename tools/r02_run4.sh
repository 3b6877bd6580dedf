#!/bin/bash
# round 2: A/B on one box: default build / machine LICM on / node prefetch; two repetitions each
set -o pipefail
O=gpurun_out/r02d
mkdir -p $O
WPT_LIB_DIR=lib_pf timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "variants or scheduler or storage_order or full_size_config_2 or sponza_like_textures" > $O/pytest_pf.log 2>&1
echo "pytest prefetch rc $?"; tail -2 $O/pytest_pf.log
for rep in 1 2; do
for lib in lib lib_licm lib_pf; do
  for wl in cornell_1024x1024_1024spp_ggx_glass sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp; do
    WPT_LIB_DIR=$lib timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${lib}_${wl}_$rep.json 2> $O/bench_${lib}_${wl}_$rep.err
    echo "$rep $lib $wl rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_${lib}_${wl}_$rep.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
  done
done
done
