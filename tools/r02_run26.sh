#!/bin/bash
# two passes again: which way round the order runs, and what each launch takes
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02ag
mkdir -p $O
timeout -k 10 120 python - > $O/equal.txt 2>&1 <<'PY' || exit 1
import numpy as np, sys
sys.path.insert(0, ".")
from wurblpt_amd import device, host
sc = host.cornell(1024, 640, 1, 2)
out = {}
for v in (0x10, 0x40):
    device.lib().wpt_set_launch_config(0, v)
    ds = device.DeviceScene(sc)
    out[v], _ = ds.render(8)
    ds.check()
device.lib().wpt_set_launch_config(0, 0)
print("two passes equal one launch:", np.array_equal(out[0x10].view(np.uint32), out[0x40].view(np.uint32)))
PY
cat $O/equal.txt | tail -1
timeout -k 10 300 python bench.py --variant 0 --no-cpu-baseline > $O/bench_co_0.json 2> $O/bench_co_0.err || exit 1
echo "cornell one launch: $(python -c "import json; d=json.load(open('$O/bench_co_0.json')); print(round(d['value'],1))")"
timeout -k 10 300 python bench.py --variant 64 --no-cpu-baseline > $O/bench_co_64.json 2> $O/bench_co_64.err || exit 1
echo "cornell two passes, longest first: $(python -c "import json; d=json.load(open('$O/bench_co_64.json')); print(round(d['value'],1))")"
WPT_ORDER_ASCENDING=1 timeout -k 10 300 python bench.py --variant 64 --no-cpu-baseline > $O/bench_co_64a.json 2> $O/bench_co_64a.err || exit 1
echo "cornell two passes, shortest first: $(python -c "import json; d=json.load(open('$O/bench_co_64a.json')); print(round(d['value'],1))")"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 bench.py --variant 64 --no-cpu-baseline > $O/trace.log 2>&1 || exit 1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r02ag/trace/t_kernel_trace.csv")))
for r in rows:
    n = r["Kernel_Name"]
    if "wpt_pathtrace" in n or "order" in n:
        print(n.split("(")[0][-60:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms, grid", r["Grid_Size_X"])
PY
