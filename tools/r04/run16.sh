#!/bin/bash
# round 4, sixteenth GPU call: light-ray ends served inside the walk of the kernels that fetch the scene from HBM: 4 / 6 / 8 / 12 lanes
set -o pipefail
O=gpurun_out/r04t
mkdir -p $O
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['library']['path'])"
}
S="--workload sponza_like_1920x1080_256spp_envmap_is"
for v in lib lib_neewalk4 lib_neewalk6 lib_neewalk8 lib_neewalk12 lib; do b sponza_${v}_$RANDOM $v "$S"; done
M="--workload measured_like_3840x2160_529spp_rgl --samples-sqrt 4 --wavefront 2"
for v in lib lib_neewalk8; do b measured_single_$v $v "$M"; done
