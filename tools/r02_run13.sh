#!/bin/bash
set -o pipefail
O=gpurun_out/r02p
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cat $O/bench_default.json
(echo "# python tools/fuzz_parity.py 500 2026 on one MI355X (gpurun), final tree of round 2: seeded random scenes of every family, GPU (counting and product kernel) vs CPU restatement, frames, work counters and ground truth arrays bit for bit"; timeout -k 10 1000 python tools/fuzz_parity.py 500 2026 2>&1 | grep -v "bounding volume" | tail -4) > $O/fuzz.txt; tail -2 $O/fuzz.txt
