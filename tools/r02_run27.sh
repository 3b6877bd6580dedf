#!/bin/bash
# two passes, order by tiles
set -o pipefail
O=gpurun_out/r02ah
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
tail -1 $O/equal.txt
for V in 0 64; do
timeout -k 10 300 python bench.py --variant $V --no-cpu-baseline > $O/bench_co_$V.json 2> $O/bench_co_$V.err || exit 1
echo "cornell variant $V: $(python -c "import json; d=json.load(open('$O/bench_co_$V.json')); print(round(d['value'],1))")"
done
for V in 0 64; do
timeout -k 10 300 python bench.py --variant $V --workload sponza_like_1920x1080_256spp_envmap_is --no-cpu-baseline > $O/bench_sp_$V.json 2> $O/bench_sp_$V.err || exit 1
echo "sponza variant $V: $(python -c "import json; d=json.load(open('$O/bench_sp_$V.json')); print(round(d['value'],1))")"
done
for V in 0 64; do
timeout -k 10 600 python bench.py --variant $V --workload courtyard_like_10M_1920x1080_121spp --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cy_$V.json 2> $O/bench_cy_$V.err || exit 1
echo "courtyard variant $V: $(python -c "import json; d=json.load(open('$O/bench_cy_$V.json')); print(round(d['value'],1))")"
done
