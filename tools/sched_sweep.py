"""usage (GPU box): python tools/sched_sweep.py [cornell|sponza] [samples_sqrt] [variant byte 0]
Times the frame for a grid of scheduler settings in one process (wpt_set_launch_config word: byte 1 = leave eighths + 1,
byte 2 = lanes a long round needs + 1, byte 3 = leaf bias); results never depend on them, only the time does."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurblpt_amd import device, host

what = sys.argv[1] if len(sys.argv) > 1 else "cornell"
ssqrt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
low = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
sc = host.cornell(1024, 1024, 1, 2) if what == "cornell" else host.sponza_like(1920, 1080)
ds = device.DeviceScene(sc)
frame = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
params = host.default_params()


def run(word):
    device.lib().wpt_set_launch_config(0, word)
    ds.render_block_into(frame, ssqrt, None, params, None, stream)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        ds.render_block_into(frame, ssqrt, None, params, None, stream)
        e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return sc.width * sc.height * ssqrt * ssqrt / best / 1e3


base = run(low)
print("default: %.1f Msamples/s" % base, flush=True)
results = []
leaves = (1, 2, 3) if what == "cornell" else (2, 3, 4)
for leave in leaves:
    for heavy in ((6, 8, 12, 16, 20, 24, 32) if what == "cornell" else (4, 8, 12, 16, 24)):
        for bias in ((8, 12, 16, 24, 32, 48) if what == "cornell" else (16, 32, 48)):
            word = low | ((leave + 1) << 8) | ((heavy + 1) << 16) | (bias << 24)
            r = run(word)
            results.append((r, leave, heavy, bias))
            print("leave %d heavy %2d bias %2d: %.1f" % (leave, heavy, bias, r), flush=True)
results.sort(reverse=True)
print("best:", results[:8])
device.lib().wpt_set_launch_config(0, 0)
