import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from wurblpt_amd import device, host
sc = host.cornell(1024, 1024, 1, 2)
ds = device.DeviceScene(sc)
w = 1024
rows = 16
costs = []
for b in range(1024 // rows):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    frame = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
    ds.render_block_into(frame, 8, (b * rows * w, rows * w), None, None, torch.cuda.current_stream())
    torch.cuda.synchronize()
    e0.record()
    ds.render_block_into(frame, 16, (b * rows * w, rows * w), None, None, torch.cuda.current_stream())
    e1.record()
    torch.cuda.synchronize()
    costs.append(e0.elapsed_time(e1))
c = np.array(costs)
print("strip ms min/mean/max", c.min(), c.mean(), c.max())
print(np.round(c, 2).tolist())
N = 8
contig = [c[r * 8:(r + 1) * 8].sum() for r in range(N)]
inter = [c[r::N].sum() for r in range(N)]
print("contiguous max/mean", max(contig) / np.mean(contig), "interleaved max/mean", max(inter) / np.mean(inter))
rng = np.random.default_rng(0)
worst = []
for t in range(1000):
    p = rng.permutation(64)
    loads = [c[p[r * 8:(r + 1) * 8]].sum() for r in range(N)]
    worst.append(max(loads) / np.mean(loads))
print("random assignment max/mean: median %.3f  90%% %.3f" % (np.median(worst), np.quantile(worst, 0.9)))
