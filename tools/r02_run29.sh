#!/bin/bash
# BVH in blocks of levels again, now that the 10 M triangle scene is bound by HBM bandwidth: rate and fetched bytes
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02aj
mkdir -p $O
Y="--workload courtyard_like_10M_1920x1080_121spp --steps 1 --warmup 1 --no-cpu-baseline"
for L in 0 2 3; do
  export WPT_NODE_BLOCK_LEVELS=$L
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_L$L -o pmc --output-format csv -- python3 bench.py $Y > $O/bench_L$L.log 2>&1 || exit 1
  python - <<PY
import csv, json
rows = [r for r in csv.DictReader(open("$O/pmc_L$L/pmc_counter_collection.csv")) if "255u, false, false" in r["Kernel_Name"]]
fetch = sum(float(r["Counter_Value"]) for r in rows) / 2 * 2048 / 1e12      # two frames; KiB -> bytes, x 2 (gfx950)
line = [l for l in open("$O/bench_L$L.log") if l.startswith('{"metric"')][-1]
print("block levels $L: %.1f Msamples/s, %.2f TB fetched per frame" % (json.loads(line)["value"], fetch))
PY
done
