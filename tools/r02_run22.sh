#!/bin/bash
set -o pipefail
O=gpurun_out/r02ac
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pixel_pool or variants or scheduler" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co.json 2> $O/bench_co.err || exit 1
echo "cornell: $(python -c "import json; d=json.load(open('$O/bench_co.json')); print(round(d['value'],1))")"
timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp.json 2> $O/bench_sp.err || exit 1
echo "sponza: $(python -c "import json; d=json.load(open('$O/bench_sp.json')); print(round(d['value'],1))")"
timeout -k 10 600 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cy.json 2> $O/bench_cy.err || exit 1
echo "courtyard: $(python -c "import json; d=json.load(open('$O/bench_cy.json')); print(round(d['value'],1))")"
timeout -k 10 200 python tools/size_scan.py > $O/size_scan.txt 2>&1 || exit 1
grep spp $O/size_scan.txt
