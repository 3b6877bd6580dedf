import os, sys
sys.path.insert(0, os.getcwd())
import torch
from wurblpt_amd import device, host
sc = host.sponza_like(1920, 1080)
ds = device.DeviceScene(sc)
frame = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
params = host.default_params()
def run(word):
    device.lib().wpt_set_launch_config(0, word)
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); ds.render_block_into(frame, 8, None, params, None, stream); e1.record(stream)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return sc.width * sc.height * 64 / best / 1e3
print("library", device.lib_path(), flush=True)
print("default %.1f" % run(0), flush=True)
for leave, heavy, bias in ((3,8,16),(3,8,8),(3,8,64),(3,8,128),(2,8,32),(4,8,32),(3,4,32),(3,16,32),(4,16,64)):
    print("leave %d heavy %2d bias %3d: %.1f" % (leave, heavy, bias, run(((leave+1)<<8)|((heavy+1)<<16)|(bias<<24))), flush=True)
