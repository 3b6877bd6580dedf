#!/bin/bash
# round 4, second GPU call: the whole GPU suite on the changed kernels (node record as pairs, NaN test per ray, wide walk with its
# stack of 96, the RCCL path at world size 1), config 4 with and without the wide walk, Cornell against round 3's library on the
# same box, the scheduler's statistics for the instruction budget, and the default bench line with its three secondaries
# (r2tree / r3tree: git worktree add -f r2tree 4383e9e; make -C r2tree/wurblpt_amd/csrc -- and r3tree at 5178f84 with its library;
# both were inside the repository for the call only, so that they travelled to the GPU box, and are removed again)
set -o pipefail
O=gpurun_out/r04b
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
WL=courtyard_like_10M_1920x1080_121spp
for v in "" "--wide-walk" "" "--wide-walk"; do
  n=$(echo "x$v" | tr -d ' -')_$RANDOM
  timeout -k 10 400 python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline $v > $O/c4_$n.json 2> $O/c4_$n.err
  python -c "import json; d=json.load(open('$O/c4_$n.json')); print('courtyard $v', round(d['value'],2), round(d['ms_per_step'],1), d['roofline'].get('kernel'))"
done
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 3 > $O/c2_new.json 2> $O/c2_new.err
python -c "import json; d=json.load(open('$O/c2_new.json')); print('cornell new', round(d['value'],2), round(d['ms_per_step'],1))"
( cd r3tree && timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 3 ) > $O/c2_r3.json 2> $O/c2_r3.err
python -c "import json; d=json.load(open('$O/c2_r3.json')); print('cornell r3', round(d['value'],2), round(d['ms_per_step'],1))"
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 3 > $O/c2_new2.json 2> $O/c2_new2.err
python -c "import json; d=json.load(open('$O/c2_new2.json')); print('cornell new', round(d['value'],2), round(d['ms_per_step'],1))"
timeout -k 10 300 python tools/sched_stats.py --json $O/sched_cornell.json > $O/sched_cornell.txt 2>&1; tail -25 $O/sched_cornell.txt
timeout -k 10 300 python tools/sched_stats.py 0 sponza --json $O/sched_sponza.json > $O/sched_sponza.txt 2>&1; tail -12 $O/sched_sponza.txt
/usr/bin/time -v timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?"; grep -E "Elapsed|Maximum resident" $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print('primary', round(d['value'],1), d['parity']['bits_differ'], d['roofline']['frac'])
for s in d['secondary']: print(s['workload'], round(s['value'],1), round(s['ms_per_step'],1), s['parity']['bits_differ'], round(s['roofline']['frac'],3), s['roofline']['kernel'], s['cpu_baseline']['value'])
"
