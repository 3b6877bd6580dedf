#!/bin/bash
# round 4, fifteenth GPU call: the end of a light ray served inside the walk once 8 / 16 / 24 lanes wait for it (variant libraries)
# against the product (they wait for the long round): Cornell, Sponza-class, 10 M triangles; parity of one variant first
set -o pipefail
O=gpurun_out/r04s
mkdir -p $O
WPT_LIB_DIR=lib_neewalk8 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cornell or sponza_like or sphere or full_size or fuzz_parity_over" > $O/pytest_neewalk8.log 2>&1; echo "pytest (neewalk8) rc $?"; tail -2 $O/pytest_neewalk8.log
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['library']['path'])"
}
for v in lib lib_neewalk8 lib_neewalk16 lib_neewalk24 lib; do b cornell_${v}_$RANDOM $v "--no-secondary"; done
S="--workload sponza_like_1920x1080_256spp_envmap_is"
for v in lib lib_neewalk8 lib_neewalk16; do b sponza_$v $v "$S"; done
Y="--workload courtyard_like_10M_1920x1080_121spp"
for v in lib lib_neewalk8; do b courtyard_$v $v "$Y"; done
