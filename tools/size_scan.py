"""Throughput of the Cornell GGX+glass frame at equal total samples but different frame sizes
(number of workgroup rounds): shows the cost of the end-of-frame tail.  GPU box."""
import sys, time
sys.path.insert(0, ".")
import torch
torch.zeros(1, device="cuda")
from wurblpt_amd import host, device
for w, s in ((512, 32), (1024, 16), (2048, 8), (4096, 4)):
    sc = host.cornell(w, w, 1, 2)
    ds = device.DeviceScene(sc)
    frame = torch.zeros((w, w, 3), dtype=torch.float32, device="cuda")
    ds.render_block_into(frame, s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        ds.render_block_into(frame, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("%dx%d x %d spp: %.1f Msamples/s (%.1f ms)" % (w, w, s * s, w * w * s * s / dt / 1e6, dt * 1e3), flush=True)
