"""usage (GPU box): python tools/two_pass_check.py [rounds [seed]]
Seeded random scenes of every family at frame sizes and sample counts where a launch takes its pixels from the pool and
(scenes fetched from HBM) renders the frame in two passes: the frame must equal, bit for bit, the one of a launch with one
lane per pixel (variant bit 0x10), in one launch and as two ranks' interleaved bands.  GPU against GPU: the one-lane-per-pixel
launch is what the parity tests and tools/fuzz_parity.py hold against the CPU restatement."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wurblpt_amd import device, host

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
bad = done = 0
t_start = time.time()


def frame_of(sc, s, p, variant):
    device.lib().wpt_set_launch_config(0, variant)
    try:
        ds = device.DeviceScene(sc)
        got, _ = ds.render(s, params=p)
        ds.check()
        return ds, got
    finally:
        device.lib().wpt_set_launch_config(0, 0)


def check(label, sc, s, p):
    global bad, done
    _, ref = frame_of(sc, s, p, 0x10)
    ds, got = frame_of(sc, s, p, 0)
    n1 = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    h_, w_ = got.shape[:2]
    band_rows = int(rng.choice([8, 16, 5]))
    total = np.zeros_like(ref)
    for rank in range(2):
        fr = torch.zeros((h_, w_, 3), dtype=torch.float32, device="cuda")
        ds.render_bands_into(fr, s, band_rows, rank, 2, params=p, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        total += fr.cpu().numpy()
    n2 = int((total.view(np.uint32) != ref.view(np.uint32)).sum())
    done += 1
    if n1 or n2:
        bad += 1
        print("MISMATCH %s: %d values differ in one launch, %d as bands" % (label, n1, n2), flush=True)


for r in range(rounds):
    seed = int(rng.integers(1, 1 << 30))
    w, h = 8 * int(rng.integers(120, 230)), 8 * int(rng.integers(90, 140))   # 0.7 to 2 M pixels
    s = int(rng.integers(8, 11))
    p = host.default_params()
    p.max_path_components = int(rng.choice([3, 8, 128]))
    p.rr_threshold = float(rng.choice([1.0, 0.5]))
    check("triangles seed %d" % seed, host.random_triangles(int(rng.integers(1000, 3000)), seed, with_texcoords=bool(rng.integers(0, 2)), width=w, height=h,
                                                         aperture=float(rng.choice([0.0, 0.05]))), s, p)
    check("sponza-like seed %d" % seed, host.sponza_like(w, h, seed=seed % 1000 + 1, detail=0.03, tex_size=16, env_width=32,
                                                        importance_n=int(rng.choice([0, 8, 16]))), s, p)
    check("courtyard-like seed %d" % seed, host.courtyard_like(w, h, seed=seed % 1000 + 1, triangles=int(rng.integers(2000, 20000)), tex_size=16), s, p)
    check("measured-like seed %d" % seed, host.measured_like(w, h, host.rgl_fixture("iso"), host.rgl_fixture("aniso"), seed=seed % 1000 + 1, detail=0.03,
                                                            tex_size=16, env_width=32, importance_n=8), s, p)
    t0 = float(rng.random())
    p.t0, p.t1 = t0, t0 + float(rng.random()) * (1.2 - t0)
    check("animated variant %d" % (r % 16), host.animated(w, h, int(r % 16), p.t0, p.t1), s, p)
    p.t0 = p.t1 = 0.0
    check("spheres variant %d" % (r % 5), host.spheres(w, h, r % 5), s, p)
    check("cornell", host.cornell(w, h, int(rng.integers(0, 2)), int(rng.choice([0, 2]))), s, p)
    print("%d scenes, %d mismatches, %.0f s" % (done, bad, time.time() - t_start), flush=True)
print("two-pass check: %d scenes, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
