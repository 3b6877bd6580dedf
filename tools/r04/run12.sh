#!/bin/bash
# round 4, twelfth GPU call: the wavefront form with a hit's light ray traced beside the path's continuation: parity (GPU suite's
# wavefront tests, tools/wf_check.py campaign), then rates and launch counts on the measured-BRDF frame
set -o pipefail
O=gpurun_out/r04m
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "wavefront or measured or rgl or config_5 or environment_light" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 300 python tools/wf_check.py parity 48 > $O/wavefront_parity.txt 2>&1; echo "wf rc $?"; tail -2 $O/wavefront_parity.txt
b() { # label, args
  timeout -k 10 600 python bench.py --no-cpu-baseline --workload measured_like_3840x2160_529spp_rgl $2 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'], d['roofline']['kernel_launches_per_launch'])"
}
b wf16_a "--samples-sqrt 4 --steps 3 --warmup 1"
b wf16_b "--samples-sqrt 4 --steps 3 --warmup 1"
b wf529 "--steps 1 --warmup 1"
timeout -k 10 300 python bench.py --no-cpu-baseline --workload sponza_like_1920x1080_256spp_envmap_is --samples-sqrt 8 --steps 3 --wavefront 1 > $O/sponza_wf.json 2> $O/sponza_wf.err
python -c "import json; d=json.load(open('$O/sponza_wf.json')); print('sponza 64 spp wavefront', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'])"
