"""usage (GPU box): python tools/wf_threshold_probe.py [samples_sqrt]
Where the wavefront form starts to pay for scenes with measured BRDFs: blocks of 2^18 .. 2^23 pixels of the Bistro-class frame,
wavefront form (wpt_set_wavefront mode 1) against the single kernel (mode 2), same pixels; the library's own choice is the
threshold in wpt_capi.hip (wfAuto)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wurblpt_amd import device, host

s = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = dict(bench.WORKLOADS["measured_like_3840x2160_529spp_rgl"])
sc = bench.build_scene(w)
ds = device.DeviceScene(sc)
width, height = w["width"], w["height"]
frame = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
p = host.default_params()
L = device.lib()
print("pixels      single kernel   wavefront   (Msamples/s)", flush=True)
for rows in (64, 128, 256, 272, 512, 1080, 2160):
    size = rows * width
    rates = []
    for mode in (2, 1):
        L.wpt_set_wavefront(mode, 0, 0, 0)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            ds.render_block_into(frame, s, (0, size), p, None, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        rates.append(size * s * s / best / 1e6)
    L.wpt_set_wavefront(0, 0, 0, 0)
    print("%9d   %10.1f   %10.1f" % (size, rates[0], rates[1]), flush=True)
