#!/bin/bash
# round 4, eighth GPU call: what the driver runs at round end, on the final tree -- GPU tests, smoke, the default bench line -- and
# the measured-BRDF shade kernel built for two waves per SIMD beside the product's three
set -o pipefail
O=gpurun_out/r04h
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.txt
T0=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print('primary', round(d['value'],1), d['parity']['bits_differ'], round(d['roofline']['frac'],4), d['roofline'].get('pmc_matches_binary'))
for s in d['secondary']: print(s['workload'], round(s['value'],1), round(s['ms_per_step'],1), s['parity']['bits_differ'], round(s['roofline']['frac'],3), s['roofline']['kernel'], round(s['cpu_baseline']['value'],2), s['roofline']['counted_on'])
"
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 600 python bench.py --no-cpu-baseline --workload measured_like_3840x2160_529spp_rgl $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'], d['library']['path'])"
}
for v in lib lib_w2 lib lib_w2; do b wf16_${v}_$RANDOM $v "--samples-sqrt 4 --steps 3 --warmup 1"; done
