"""usage (GPU box): python tools/redeal_cost.py [samples_sqrt]
What a deal costs and what it gives (Cornell frame): the re-dealing kernel with every path left in its lane
(WPT_REDEAL_IDENTITY: all barriers and LDS traffic, no regrouping) next to the real thing, per cadence."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurblpt_amd import device, host

ssqrt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = host.cornell(1024, 1024, 1, 2)
ds = device.DeviceScene(sc)
frame = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
params = host.default_params()


def run(every, identity):
    os.environ.pop("WPT_REDEAL", None)
    os.environ.pop("WPT_REDEAL_IDENTITY", None)
    if every:
        os.environ["WPT_REDEAL"] = str(every)
    if identity:
        os.environ["WPT_REDEAL_IDENTITY"] = "1"
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        ds.render_block_into(frame, ssqrt, None, params, None, stream)
        e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return sc.width * sc.height * ssqrt * ssqrt / best / 1e3


print("plain kernel, four workgroups per compute unit: %.1f Msamples/s" % run(0, False), flush=True)
for every in (1, 2, 4, 8, 10000):
    print("every %5d: paths stay %.1f, paths sorted %.1f" % (every, run(every, True), run(every, False)), flush=True)
