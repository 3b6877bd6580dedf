#!/bin/bash
# round 4, fourteenth GPU call: one group by default: the measured-BRDF frame in full, the wavefront tests, and the PMC passes of that frame
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04o
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "wavefront or measured or rgl or config_5" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout -k 10 600 python bench.py --no-cpu-baseline --workload measured_like_3840x2160_529spp_rgl --steps 2 --warmup 1 > $O/wf529.json 2> $O/wf529.err
python -c "import json; d=json.load(open('$O/wf529.json')); print('wf529', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['kernel'], d['roofline']['kernel_launches_per_launch'])"
for P in "WRITE_SIZE" "FETCH_SIZE"; do
  timeout -k 10 500 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_$P -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --workload measured_like_3840x2160_529spp_rgl --samples-sqrt 8 --steps 1 --warmup 0 > $O/pmc_$P.log 2>&1 || { echo "FAILED $P"; exit 1; }
done
python - <<'PY'
import csv, collections
for P in ("WRITE_SIZE", "FETCH_SIZE"):
    tot = collections.defaultdict(float)
    for r in csv.DictReader(open("gpurun_out/r04o/pmc_%s/pmc_counter_collection.csv" % P)):
        k = r["Kernel_Name"]
        if "wf_" in k:
            tot[k.split("(")[0][-30:]] += float(r["Counter_Value"])
    print(P, "KiB per frame of 64 spp:", dict(tot), "sum", sum(tot.values()), "-> per 529-spp frame x 8.27:", sum(tot.values()) * 529 / 64)
PY
