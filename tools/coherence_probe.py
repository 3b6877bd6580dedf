"""usage (GPU box): python tools/coherence_probe.py [samples_sqrt [cornell]]
How fast do walks run when the rays of a wave belong together?  The Sponza-class frame with paths cut after 2, 3, 4, ... components
(Parameters::maxPathComponents): with 2 there are camera rays and the light rays from their hits only -- neighbouring pixels,
neighbouring origins --, every further component adds rays that have been scattered once more.  Node visits per second
(visits from a counting launch with the product's walks, wpt_set_walk(WPT_WALK_COUNT_PRODUCT); time from the rendering launch) per depth
says what sorting rays could at best recover of the difference between the first and the last line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurblpt_amd import device, host

ssqrt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# "cornell": the same question where no memory is in the way (scene in LDS): what is lost to lanes that part ways
sc = host.cornell(1024, 1024, 1, 2) if len(sys.argv) > 2 and sys.argv[2] == "cornell" else host.sponza_like(1920, 1080, seed=1)
ds = device.DeviceScene(sc)
frame = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
names = ("samples", "rays", "node_visits", "leaf_tests", "pdf_tests", "scatters")
print("components  rays/sample  visits/sample  ms/frame  G visits/s  Msamples/s", flush=True)
for depth in (2, 3, 4, 6, 10, 0):
    params = host.default_params()
    if depth:
        params.max_path_components = depth
    counters = torch.zeros(6, dtype=torch.int64, device="cuda")
    device.lib().wpt_set_walk(device.WALK_COUNT_PRODUCT)
    ds.render_block_into(frame, ssqrt, None, params, counters, stream)
    torch.cuda.synchronize()
    device.lib().wpt_set_walk(0)
    cnt = dict(zip(names, [int(x) for x in counters.cpu().tolist()]))
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        ds.render_block_into(frame, ssqrt, None, params, None, stream)
        e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    n = float(cnt["samples"])
    print("%10s  %11.2f  %13.1f  %8.1f  %10.1f  %11.1f" % (depth or "default (%d)" % host.default_params().max_path_components, cnt["rays"] / n, cnt["node_visits"] / n, best,
                                                     cnt["node_visits"] / best / 1e6, n / best / 1e3), flush=True)
