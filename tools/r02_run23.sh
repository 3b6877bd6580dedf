#!/bin/bash
# the other configurations and the two-rank rehearsal on the tree with the pixel pool
set -o pipefail
O=gpurun_out/r02ad
mkdir -p $O
WPT_BENCH_DEVICE=0 WPT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 2 --warmup 1 --verify > $O/bench_cornell_n2.json 2> $O/bench_cornell_n2.err
echo "cornell N=2 (gloo, one GPU) rc $? $(grep '^{' $O/bench_cornell_n2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d.get('frame_equals_single_launch'))")"
timeout -k 10 300 python bench.py --workload cornell_256x256_64spp_lambertian --steps 20 --warmup 3 > $O/bench_c1.json 2> $O/bench_c1.err || exit 1
echo "config 1: $(python -c "import json; d=json.load(open('$O/bench_c1.json')); print(round(d['value'],1))")"
timeout -k 10 600 python bench.py --workload measured_like_3840x2160_529spp_rgl --steps 1 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err || exit 1
echo "config 5: $(python -c "import json; d=json.load(open('$O/bench_c5.json')); print(round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])")"
