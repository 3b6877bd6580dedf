#!/bin/bash
set -o pipefail
O=gpurun_out/r02h
mkdir -p $O
for st in 2 3 4 6 8; do
  WPT_STEPS=$st timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "variants or full_size_config_2" > $O/pytest_$st.log 2>&1; echo "pytest steps $st rc $?"
  WPT_STEPS=$st timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cornell_s$st.json 2> $O/bench_cornell_s$st.err
  echo "cornell steps $st rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_cornell_s$st.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
done
for st in 2 3 4; do
  WPT_STEPS=$st timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "sponza_like_textures or courtyard_like or storage_order" > $O/pytest_full_$st.log 2>&1; echo "pytest full steps $st rc $?"
  for wl in sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp; do
    WPT_STEPS=$st timeout -k 10 500 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${wl}_s$st.json 2> $O/bench_${wl}_s$st.err
    echo "$wl steps $st rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_${wl}_s$st.json')); print(round(d['value'],1), round(d['ms_per_step'],1))")"
  done
done
