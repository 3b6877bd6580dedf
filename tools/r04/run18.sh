#!/bin/bash
# round 4, eighteenth GPU call: the traversal round also ends when 16 / 24 / 32 / 40 lanes wait for their shading (variant libraries)
set -o pipefail
O=gpurun_out/r04y
mkdir -p $O
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['library']['path'])"
}
S="--workload sponza_like_1920x1080_256spp_envmap_is"
for v in lib lib_sb16 lib_sb24 lib_sb32 lib_sb40 lib; do b sponza_${v}_$RANDOM $v "$S"; done
Y="--workload courtyard_like_10M_1920x1080_121spp"
for v in lib lib_sb24 lib_sb32; do b courtyard_$v $v "$Y"; done
for v in lib lib_sbl32; do b cornell_$v $v "--no-secondary"; done
