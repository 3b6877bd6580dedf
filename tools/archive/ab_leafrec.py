"""usage (GPU box): python tools/ab_leafrec.py workload samples_sqrt
Frame time of a workload with the triangles' corners behind their leaf nodes (wpt_set_top_nodes bit 31) and without (default),
same process, same kernels; frames must be equal bit for bit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from wurblpt_amd import device, host

name, s = sys.argv[1], int(sys.argv[2])
w = dict(bench.WORKLOADS[name])
sc = bench.build_scene(w)
L = device.lib()
frames = {}
for label, word in (("leaf records", 65536 | 0x80000000), ("triangle array only", 65536), ("leaf records", 65536 | 0x80000000)):
    L.wpt_set_top_nodes(word)
    ds = device.DeviceScene(sc)
    frame = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
    best = None
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ds.render_block_into(frame, s, None, host.default_params(), None, torch.cuda.current_stream())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    frames[label] = frame.cpu().numpy()
    print("%-22s %8.1f ms  %7.2f Msamples/s" % (label, best * 1e3, w["width"] * w["height"] * s * s / best / 1e6), flush=True)
    del ds
L.wpt_set_top_nodes(65536)
print("frames equal:", np.array_equal(frames["leaf records"].view(np.uint32), frames["triangle array only"].view(np.uint32)))
