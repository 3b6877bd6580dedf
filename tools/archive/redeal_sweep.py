"""usage (GPU box): python tools/redeal_sweep.py [samples_sqrt]
The Cornell frame with the kernel that re-deals a workgroup's paths to its lanes (WPT_REDEAL=n: at every n-th look at the lane
counts), over the scheduler's settings (wpt_set_launch_config: byte 1 = leave eighths + 1, byte 2 = lanes a long round needs
+ 1, byte 3 = leaf bias).  Results never depend on any of it: the frames are compared bit for bit with the plain kernel's."""
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


def run(word, every):
    if every:
        os.environ["WPT_REDEAL"] = str(every)
    else:
        os.environ.pop("WPT_REDEAL", None)
    device.lib().wpt_set_launch_config(0, word)
    best = 1e30
    for _ in range(2):
        frame.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        ds.render_block_into(frame, ssqrt, None, params, None, stream)
        e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return sc.width * sc.height * ssqrt * ssqrt / best / 1e3, frame.clone()


base, ref = run(0, 0)
print("plain kernel: %.1f Msamples/s" % base, flush=True)
results = []
for every in (1, 2, 3, 4, 8):
    for leave in (1, 2, 4, 6):
        for heavy in (8, 16, 32, 48):
            for bias in (8, 16, 32):
                word = ((leave + 1) << 8) | ((heavy + 1) << 16) | (bias << 24)
                r, f = run(word, every)
                same = bool(torch.equal(f.view(torch.int32), ref.view(torch.int32)))
                results.append((r, every, leave, heavy, bias))
                print("every %d leave %d heavy %2d bias %2d: %.1f%s" % (every, leave, heavy, bias, r, "" if same else "  FRAME DIFFERS"), flush=True)
results.sort(reverse=True)
print("best:", results[:10])
device.lib().wpt_set_launch_config(0, 0)
