#!/bin/bash
# round 4, eleventh GPU call: the parity campaigns on the final library -- random scenes of every family against the CPU restatement
# (binary walk, then uploaded with the wide form), wavefront kernels against the single kernel, pool / two-pass launches against one
# lane per pixel -- then the default bench line once more (the record of the final library) and the GPU suite
set -o pipefail
O=gpurun_out/r04k
mkdir -p $O
timeout -k 10 500 python tools/fuzz_parity.py 60 2026 > $O/fuzz_parity.txt 2>&1; echo "fuzz rc $?"; tail -1 $O/fuzz_parity.txt
timeout -k 10 300 python tools/fuzz_parity.py 30 77 1 --wide > $O/fuzz_parity_wide.txt 2>&1; echo "fuzz wide rc $?"; tail -1 $O/fuzz_parity_wide.txt
timeout -k 10 300 python tools/wf_check.py parity 48 > $O/wavefront_parity.txt 2>&1; echo "wf rc $?"; tail -1 $O/wavefront_parity.txt
timeout -k 10 300 python tools/two_pass_check.py 24 > $O/two_pass_check.txt 2>&1; echo "two-pass rc $?"; tail -1 $O/two_pass_check.txt
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.txt
T0=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print('primary', round(d['value'],1), d['parity']['bits_differ'], round(d['roofline']['frac'],4), d['roofline'].get('pmc_matches_binary'), d['library']['sha256'])
for s in d['secondary']: print(s['workload'], round(s['value'],1), round(s['ms_per_step'],1), s['parity']['bits_differ'], round(s['roofline']['frac'],3), s['roofline']['kernel'], round(s['cpu_baseline']['value'],2), s['roofline'].get('pmc_matches_binary'))
"
