#!/bin/bash
# what the driver runs at round end, on the final tree: GPU tests, smoke, the default bench line; then the other configurations
set -o pipefail
O=gpurun_out/r02_final
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.txt
timeout -k 10 600 python bench.py > $O/bench_cornell.json 2> $O/bench_cornell.err; echo "bench rc $?"
python -c "import json; d=json.load(open('$O/bench_cornell.json')); print(d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['cpu_baseline']['value'])"
for wl in sponza_like_1920x1080_256spp_envmap_is courtyard_like_10M_1920x1080_121spp; do
  timeout -k 10 900 python bench.py --workload $wl --steps 2 --warmup 1 > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "import json; d=json.load(open('$O/bench_$wl.json')); print('$wl', round(d['value'],1), d['roofline']['bound'], round(d['roofline']['frac'],3), round(d['roofline'].get('hbm_gbps_from_traffic',0)), d['cpu_baseline']['value'])"
done
