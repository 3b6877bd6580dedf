#!/bin/bash
# round 4, sixth GPU call: measured BRDFs with the interleaved colour + luminance table: parity, then rates against the library
# of the call before (wurblpt_amd/lib_before) on the same box
set -o pipefail
O=gpurun_out/r04f
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "rgl or measured or config_5 or fuzz or wavefront" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
b() { # label, lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 600 python bench.py --no-cpu-baseline --workload measured_like_3840x2160_529spp_rgl $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'], d['library']['path'])"
}
b wf16_before lib_before "--samples-sqrt 4 --steps 3 --warmup 1"
b wf16_now lib "--samples-sqrt 4 --steps 3 --warmup 1"
b sk16_before lib_before "--samples-sqrt 4 --steps 3 --warmup 1 --wavefront 2"
b sk16_now lib "--samples-sqrt 4 --steps 3 --warmup 1 --wavefront 2"
b wf16_before2 lib_before "--samples-sqrt 4 --steps 3 --warmup 1"
b wf16_now2 lib "--samples-sqrt 4 --steps 3 --warmup 1"
b wf529_now lib "--steps 1 --warmup 1"
