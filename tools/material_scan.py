"""Throughput of the Cornell frame 1024x1024x256 spp with different object materials (GPU box):
shows what material divergence in the long round costs."""
import sys, time
sys.path.insert(0, ".")
import torch
torch.zeros(1, device="cuda")
from wurblpt_amd import host, device
for tall, short, name in ((0, 0, "all Lambertian"), (1, 0, "GGX tall box"), (0, 2, "glass short box"), (1, 2, "GGX + glass (config 2)")):
    sc = host.cornell(1024, 1024, tall, short)
    ds = device.DeviceScene(sc)
    frame = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
    ds.render_block_into(frame, 16)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        ds.render_block_into(frame, 16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("%-26s %.1f Msamples/s" % (name, 1024 * 1024 * 256 / dt / 1e6), flush=True)
