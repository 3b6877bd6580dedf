#!/bin/bash
# round 4, first GPU call: parity of the cleaned kernels and of the wide walk, then config 4 under three libraries on ONE box
# (round 2's tree at 4383e9e, round 3's at 5178f84, this tree; this tree also with the wide walk), twice over, then the other workloads
# (r2tree / r3tree: git worktree add -f r2tree 4383e9e; make -C r2tree/wurblpt_amd/csrc -- and r3tree at 5178f84 with its library;
# both were inside the repository for the call only, so that they travelled to the GPU box, and are removed again)
set -o pipefail
O=gpurun_out/r04a
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or environment_light or wavefront_kernels or cornell_frame or full_size_config_2 or sponza_like or fuzz" > $O/pytest_subset.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest_subset.log
WL=courtyard_like_10M_1920x1080_121spp
run() { # label, directory, extra args
  ( cd $2 && timeout -k 10 400 python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline $3 ) > $O/c4_$1.json 2> $O/c4_$1.err
  python -c "import json; d=json.load(open('$O/c4_$1.json')); print('$1', round(d['value'],2), round(d['ms_per_step'],1), d['roofline'].get('kernel'))"
}
run r2_a r2tree "" && run r3_a r3tree "" && run new_a . "" && run wide_a . "--wide-walk" && run r2_b r2tree "" && run r3_b r3tree "" && run new_b . "" && run wide_b . "--wide-walk"
for wl in sponza_like_1920x1080_256spp_envmap_is; do
  for v in "" "--wide-walk"; do
    timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline $v > $O/c3$v.json 2> $O/c3$v.err
    python -c "import json; d=json.load(open('$O/c3$v.json')); print('$wl $v', round(d['value'],2), round(d['ms_per_step'],1), d['roofline'].get('kernel'))"
  done
done
( cd r3tree && timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --steps 3 --warmup 1 --no-cpu-baseline ) > $O/c3_r3.json 2> $O/c3_r3.err
python -c "import json; d=json.load(open('$O/c3_r3.json')); print('sponza r3', round(d['value'],2), round(d['ms_per_step'],1))"
for v in "--wavefront 2" "--wavefront 2 --wide-walk" ""; do
  n=$(echo "$v" | tr -d ' -')
  timeout -k 10 400 python bench.py --workload measured_like_3840x2160_529spp_rgl --samples-sqrt 4 --steps 3 --warmup 1 --no-cpu-baseline $v > $O/c5_$n.json 2> $O/c5_$n.err
  python -c "import json; d=json.load(open('$O/c5_$n.json')); print('measured 16spp $v', round(d['value'],2), round(d['ms_per_step'],1), d['roofline'].get('kernel'))"
done
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 3 > $O/c2_new.json 2> $O/c2_new.err
python -c "import json; d=json.load(open('$O/c2_new.json')); print('cornell new', round(d['value'],2), round(d['ms_per_step'],1))"
( cd r3tree && timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 3 ) > $O/c2_r3.json 2> $O/c2_r3.err
python -c "import json; d=json.load(open('$O/c2_r3.json')); print('cornell r3', round(d['value'],2), round(d['ms_per_step'],1))"
