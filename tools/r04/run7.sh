#!/bin/bash
# round 4, seventh GPU call: the interleaved colour + luminance table, fused look-up: with the model as calls (product) and inlined
# (variant), against the library without the table, wavefront form and single kernel at 16 spp
set -o pipefail
O=gpurun_out/r04g
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "rgl or measured or config_5" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -2 $O/pytest.log
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 600 python bench.py --no-cpu-baseline --workload measured_like_3840x2160_529spp_rgl $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'], d['library']['path'])"
}
for v in lib_before lib lib_ilinl lib_before lib lib_ilinl; do b wf16_${v}_$RANDOM $v "--samples-sqrt 4 --steps 3 --warmup 1"; done
for v in lib_before lib; do b sk16_$v $v "--samples-sqrt 4 --steps 3 --warmup 1 --wavefront 2"; done
