#!/bin/bash
set -o pipefail
O=gpurun_out/r02ai
mkdir -p $O
for wl in sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp; do
  timeout -k 10 900 python bench.py --workload $wl --steps 2 --warmup 1 > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_$wl.json')); print('$wl', round(d['value'],1), d['ms_per_step'], d['roofline']['bound'], round(d['roofline']['frac'],3), round(d['roofline'].get('hbm_gbps_from_traffic',0)), d['cpu_baseline']['value'])"
done
timeout -k 10 600 python bench.py --workload measured_like_3840x2160_529spp_rgl --steps 1 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err || exit 1
echo "config 5: $(python -c "import json; d=json.load(open('$O/bench_c5.json')); print(round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])")"
