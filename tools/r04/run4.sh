#!/bin/bash
# round 4, fourth GPU call: GPU suite (triangle order test is new), Cornell with variant libraries (node steps per look, NEW lanes
# that wait for company, kinds that stand back), triangle order A/B on the scenes in HBM, scheduler sweeps on the new kernels, and
# the memory-pipeline counter groups, now within the counter slots of a block
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
b() { # label, env lib dir, args
  WPT_LIB_DIR=$2 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $3 > $O/$1.json 2> $O/$1.err
  python -c "import json; d=json.load(open('$O/$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['library']['path'])"
}
for v in lib lib_steps2 lib_steps4 lib_newmin8 lib_newmin16 lib; do b cornell_${v}_$RANDOM $v "--no-secondary"; done
for w in 4 8 12; do b cornell_variant$w lib "--no-secondary --variant $w"; done
S="--workload sponza_like_1920x1080_256spp_envmap_is"
for v in lib lib_newmin8 lib_newmin16; do b sponza_$v $v "$S"; done
b sponza_asgiven lib "$S --triangles-as-given"
Y="--workload courtyard_like_10M_1920x1080_121spp"
b courtyard_default lib "$Y"; b courtyard_asgiven lib "$Y --triangles-as-given"; b courtyard_default2 lib "$Y"
timeout -k 10 400 python tools/sched_sweep.py cornell > $O/sweep_cornell.txt 2>&1; sort -t: -k2 -n -r $O/sweep_cornell.txt | head -6
timeout -k 10 400 python tools/sched_sweep.py sponza 8 > $O/sweep_sponza.txt 2>&1; sort -t: -k2 -n -r $O/sweep_sponza.txt | head -6
bash tools/pmc_memory_pipeline.sh r04mp --workload sponza_like_1920x1080_256spp_envmap_is --samples-sqrt 8 --steps 1 --warmup 0 --no-cpu-baseline
python tools/pmc_sum.py r04mp wpt_pathtrace > $O/pmc_memory_pipeline.txt 2>&1; tail -40 $O/pmc_memory_pipeline.txt
