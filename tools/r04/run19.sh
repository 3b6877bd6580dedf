#!/bin/bash
# round 4, nineteenth GPU call: the wavefront trace asks for a triangle's corners when it finds the leaf (wurblpt_amd/lib) against the
# library before (lib_before): parity of the wavefront tests, measured-BRDF frame at 16 spp and in full, Sponza-class in wavefront form
set -o pipefail
O=gpurun_out/r04z
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "wavefront or measured or config_5" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 600 python bench.py --no-cpu-baseline $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'], d['library']['path'])"
}
M="--workload measured_like_3840x2160_529spp_rgl"
for v in lib_before lib lib_before lib; do b wf16_${v}_$RANDOM $v "$M --samples-sqrt 4 --steps 3 --warmup 1"; done
b wf529_lib lib "$M --steps 1 --warmup 1"
S="--workload sponza_like_1920x1080_256spp_envmap_is --samples-sqrt 8 --steps 3 --wavefront 1"
for v in lib_before lib; do b sponza_wf_$v $v "$S"; done
