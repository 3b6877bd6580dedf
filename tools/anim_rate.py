"""usage (GPU box): python tools/anim_rate.py -- throughput of the animated test scene against the same scene at rest"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurblpt_amd import device, host

W = H = 1024
S = 8
for variant, t0, t1, label in ((8, 0.0, 1.0, "moving instances + marbles + camera, exposure 0..1"), (12, 0.0, 1.0, "only the camera moves, exposure 0..1"),
                               (12, 0.3, 0.3, "the same at rest (t0 == t1), static kernel")):
    sc = host.animated(W, H, variant, t0, t1)
    ds = device.DeviceScene(sc)
    p = host.default_params()
    p.t0, p.t1 = t0, t1
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    main = torch.cuda.current_stream()
    ds.render_block_into(frame, 2, None, p, None, main)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    ds.render_block_into(frame, S, None, p, None, main)
    e1.record(main)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%-55s %8.1f ms  %7.1f Msamples/s  kernel %s" % (label, ms, W * H * S * S / ms / 1e3, device.lib().wpt_kernel_name().decode()))
