#!/bin/bash
# profile round part 2 (10 M triangles), the N = 2 start-up rehearsal, the share costs, the other configurations
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02o
mkdir -p $O
bash tools/profile_round.sh r02 2 > gpurun_out/profile_r02_part2.log 2>&1; echo "part2 rc $?"; tail -1 gpurun_out/profile_r02_part2.log
timeout -k 10 600 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_courtyard_n1.json 2> $O/bench_courtyard_n1.err
echo "courtyard N=1 build $(python -c "import json; d=json.load(open('$O/bench_courtyard_n1.json')); print(round(d['scene_build_s'],1), round(d['value'],1))")"
WPT_BENCH_DEVICE=0 WPT_BENCH_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload courtyard_like_10M_1920x1080_121spp --steps 1 --warmup 0 --verify > $O/bench_courtyard_n2.json 2> $O/bench_courtyard_n2.err
echo "courtyard N=2 (gloo, one GPU) rc $? $(grep '^{' $O/bench_courtyard_n2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['scene_build_s'],1), round(d['value'],1), d.get('frame_equals_single_launch'))")"
for n in 2 4 8; do timeout -k 10 300 python tools/share_cost.py $n > $O/share_$n.txt 2>&1; tail -1 $O/share_$n.txt; done
for wl in cornell_256x256_64spp_lambertian measured_like_3840x2160_529spp_rgl; do
  timeout -k 10 900 python bench.py --workload $wl --steps 2 --warmup 1 > $O/bench_$wl.json 2> $O/bench_$wl.err
  echo "$wl $(python -c "import json; d=json.load(open('$O/bench_$wl.json')); print(round(d['value'],1), d['cpu_baseline']['value'])")"
done
