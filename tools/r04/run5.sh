#!/bin/bash
# round 4, fifth GPU call: the measured-BRDF model inlined into its kernels (variant library) against the product, wavefront form
# and single kernel, 16 spp and the full frame
set -o pipefail
O=gpurun_out/r04e
mkdir -p $O
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 600 python bench.py --no-cpu-baseline --workload measured_like_3840x2160_529spp_rgl $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'], d['library']['path'])"
}
b wf16_lib lib "--samples-sqrt 4 --steps 3 --warmup 1"
b wf16_inline lib_rglinline "--samples-sqrt 4 --steps 3 --warmup 1"
b sk16_lib lib "--samples-sqrt 4 --steps 3 --warmup 1 --wavefront 2"
b sk16_inline lib_rglinline "--samples-sqrt 4 --steps 3 --warmup 1 --wavefront 2"
b wf16_lib2 lib "--samples-sqrt 4 --steps 3 --warmup 1"
b wf16_inline2 lib_rglinline "--samples-sqrt 4 --steps 3 --warmup 1"
