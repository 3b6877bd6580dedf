"""usage: python tools/sched_stats.py [variant-word] [sponza] [--json out.json]
Prints the wave scheduler's statistics for the Cornell GGX+glass frame (GPU box)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
json_path = None
if "--json" in sys.argv:  # for tools/instruction_budget.py
    at = sys.argv.index("--json")
    json_path = sys.argv[at + 1]
    del sys.argv[at:at + 2]
import torch
from wurblpt_amd import device, host
var = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if var:
    device.lib().wpt_set_launch_config(0, var)
sc = host.sponza_like(1920, 1080) if (len(sys.argv) > 2 and sys.argv[2] == "sponza") else host.cornell(1024, 1024, 1, 2)
ds = device.DeviceScene(sc)
stats = torch.zeros(72, dtype=torch.int64, device="cuda")
device.lib().wpt_set_scheduler_stats.argtypes = [C.c_void_p]
device.lib().wpt_set_scheduler_stats(C.c_void_p(stats.data_ptr()))
spp_sqrt = 2 if len(sys.argv) > 2 else 4
frame, cnt = ds.render(spp_sqrt, with_counters=True)
if len(sys.argv) > 3 and sys.argv[3] == "pool":
    # statistics of the ray-pool kernel: work counters from the counted launch above, scheduler
    # statistics from an uncounted launch
    stats.zero_()
    ds.render(spp_sqrt)
s = [int(x) for x in stats.cpu().tolist()]
n = cnt["samples"]
if json_path:
    import json
    json.dump({"samples": n, "counters": cnt, "sched": s[:24], "sec_wave": s[24:48], "sec_lane": s[48:72]}, open(json_path, "w"))
print("per sample:", {k: round(v / n, 3) for k, v in cnt.items()})
names = ["NODE", "LEAF", "SHADE", "NEEEND", "NEW"]
print("NODE: rounds/sample-lane %.3f iters/round %.2f avg active lanes %.1f  (lane-steps %.1f/sample)" % (s[0] * 64 / n, s[1] / max(1, s[0]), s[2] / max(1, s[1]), s[2] / n))
for i, nm in enumerate(names[1:]):
    r, l = s[3 + 2 * i], s[4 + 2 * i]
    print("%s: rounds per 64 samples %.3f avg lanes %.1f (lane-execs %.3f/sample)" % (nm, r * 64 / n, l / max(1, r), l / n))
tot_rounds = s[0] + s[3] + s[5] + s[7] + s[9]
print("scheduler rounds per 64 samples: %.2f" % (tot_rounds * 64 / n))
if sum(s[11:15]) > 0:
    tot = float(sum(s[11:15]))
    print("shader clock per block kind: traversal %.1f%%  shade %.1f%%  nee-end %.1f%%  new %.1f%%  (ticks per 64 samples: %.0f)" % (
        100 * s[11] / tot, 100 * s[12] / tot, 100 * s[13] / tot, 100 * s[14] / tot, tot * 64 / n))
    print("ticks per round: node-iter %.0f (incl. leaf iters)  shade %.0f  nee-end %.0f  new %.0f" % (
        s[11] / max(1, s[1] + s[3]), s[12] / max(1, s[5]), s[13] / max(1, s[7]), s[14] / max(1, s[9])))
if sum(s[16:24]) > 0:
    tot = float(sum(s[16:24]))
    names = ["hit record + material", "scatter", "emission", "light pdf (scattered dir)", "light sample", "light pdf (light dir)", "evaluation towards light", "env sampling / continuation"]
    print("SHADE block, lane-weighted clock per section:")
    for k, nm in enumerate(names):
        print("  %-28s %5.1f%%" % (nm, 100 * s[16 + k] / tot))
