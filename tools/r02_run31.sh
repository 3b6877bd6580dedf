#!/bin/bash
# rays of a long round start together (one evaluation of the ray's reciprocals and shear per round)
set -o pipefail
O=gpurun_out/r02an
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -1 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_co.json 2> $O/bench_co.err || exit 1
echo "cornell: $(python -c "import json; d=json.load(open('$O/bench_co.json')); print(round(d['value'],1))")"
timeout -k 10 300 python bench.py --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp.json 2> $O/bench_sp.err || exit 1
echo "sponza: $(python -c "import json; d=json.load(open('$O/bench_sp.json')); print(round(d['value'],1))")"
timeout -k 10 600 python bench.py --workload courtyard_like_10M_1920x1080_121spp --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cy.json 2> $O/bench_cy.err || exit 1
echo "courtyard: $(python -c "import json; d=json.load(open('$O/bench_cy.json')); print(round(d['value'],1))")"
