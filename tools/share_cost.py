"""usage (GPU box): python tools/share_cost.py [N] -- time of one rank's share of the Cornell frame at N ranks:
interleaved bands in one launch against the same bands as blocks on 8 streams, and what N GPUs would give."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurblpt_amd import blocks, device, host

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:
    device.lib().wpt_set_launch_config(0, int(sys.argv[2], 0))
W = H = 1024
S = 32
sc = host.cornell(W, H, 1, 2)
ds = device.DeviceScene(sc)
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
bs = blocks.plan_block_size(W * H, W, N, 8)
rows = bs // W
main = torch.cuda.current_stream()


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps):
        fn()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


full = timed(lambda: ds.render_block_into(frame, S, None, None, None, main))
for rank in (0, N // 2):
    one = timed(lambda: ds.render_bands_into(frame, S, rows, rank, N, stream=main))
    streams = [torch.cuda.Stream() for _ in range(8)]
    mine = [b for b in range(rank, (W * H) // bs, N)]

    def strips():
        for s in streams:
            s.wait_stream(main)
        for k, b in enumerate(mine):
            ds.render_block_into(frame, S, (b * bs, bs), None, None, streams[k % 8])
        for s in streams:
            main.wait_stream(s)
    many = timed(strips)
    print("N=%d rank %d: full frame %.1f ms; share in one launch %.1f ms (x%.2f of ideal), as %d strips on 8 streams %.1f ms" % (
        N, rank, full, one, one / (full / N), len(mine), many))
print("projected speedup at N=%d from the one-launch share: %.2f of %d" % (N, full / one, N))
