"""usage (GPU box): python tools/fuzz_parity.py [rounds [seed [scale]]] [--wide]
Seeded random scenes of every family through the GPU and the CPU restatement: frames, work counters and the ground truth
arrays must agree bit for bit; a random pixel block and a random split into interleaved bands must give the same frame.
--wide: scenes are uploaded with the wide form of their tree (wpt_set_walk) and product launches walk that.
Prints one line per mismatch and a summary; exit code 1 if anything differed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wurblpt_amd import device, host
from tests import oracle_loader

wide = "--wide" in sys.argv
sys.argv = [a for a in sys.argv if a != "--wide"]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
orc = oracle_loader.load("portable")
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # frames `scale` times wider and higher
bad = 0
done = 0
t_start = time.time()
if wide and not os.environ.get("WPT_FUZZ_LIST"):
    device.lib().wpt_set_walk(device.WALK_WIDE)


def check(label, sc, s, p=None, tables=False):
    global bad, done
    if os.environ.get("WPT_FUZZ_LIST"):
        # no rendering (runs without a GPU): the same draws from the generator, and what each scene was rendered with
        w_, h_ = sc.width, sc.height
        start = int(rng.integers(0, w_ * h_))
        rng.integers(1, w_ * h_ - start + 1)
        rng.integers(1, 20), rng.integers(1, 6)
        print("%s: %dx%d, samples_sqrt %d, max_path_components %s, rr_threshold %s, randomize_ray_over_pixel %s" % (
            label, w_, h_, s, getattr(p, "max_path_components", None), getattr(p, "rr_threshold", None), getattr(p, "randomize_ray_over_pixel", None)), flush=True)
        done += 1
        return
    if tables and sc.d.envmap.N > 0:
        t = orc.envmap_tables(sc)
        ds = device.DeviceScene(sc)       # the device builds its own tables at upload
        sc.set_envmap_tables(*t)
    else:
        ds = device.DeviceScene(sc)
    ref, rc = orc.render(sc, s, p)
    got, gc = ds.render(s, params=p, with_counters=True)
    got2, _ = ds.render(s, params=p)
    n1 = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    n2 = int((got2.view(np.uint32) != ref.view(np.uint32)).sum())
    # the same frame through a random block and through interleaved bands (one launch per "rank")
    import torch
    h_, w_ = got.shape[:2]
    start = int(rng.integers(0, w_ * h_))
    size = int(rng.integers(1, w_ * h_ - start + 1))
    part, _ = ds.render(s, block=(start, size), params=p)
    n4 = int((part.reshape(-1, 3)[start:start + size].view(np.uint32) != ref.reshape(-1, 3)[start:start + size].view(np.uint32)).sum())
    n4 += int(np.count_nonzero(part.reshape(-1, 3)[:start])) + int(np.count_nonzero(part.reshape(-1, 3)[start + size:]))
    band_rows, stride = int(rng.integers(1, 20)), int(rng.integers(1, 6))
    total = np.zeros_like(ref)
    for rank in range(stride):
        fr = torch.zeros((h_, w_, 3), dtype=torch.float32, device="cuda")
        ds.render_bands_into(fr, s, band_rows, rank, stride, params=p, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        total += fr.cpu().numpy()
    n4 += int((total.view(np.uint32) != ref.view(np.uint32)).sum())
    # the ground truth pass of the same scene (pixel space flow only where the camera has an image plane)
    cam = sc.camera.contents
    bits = device.GT_ALL if (cam.surround_mode == 0 and cam.stereoscopic_distance <= 0.0) else device.GT_ALL & ~((1 << 17) | (1 << 18))
    t0 = p.t0 if p is not None else 0.0
    times = (t0, max(0.0, t0 - 0.2), t0 + 0.3)
    gref = orc.ground_truth(sc, bits=bits, times=times)
    ggot = device.ground_truth(ds, bits=bits, times=times)
    n3 = sum(int((ggot[k].view(np.uint32) != gref[k].view(np.uint32)).sum()) for k in gref)
    done += 1
    if n1 or n2 or n3 or n4 or gc != rc:
        bad += 1
        print("MISMATCH %s: %d / %d values differ (counting / product kernel), counters equal: %s, ground truth values differing: %d, "
              "block / bands values differing: %d" % (label, n1, n2, gc == rc, n3, n4), flush=True)


for r in range(rounds):
    seed = int(rng.integers(1, 1 << 30))
    w, h = scale * int(rng.integers(17, 72)), scale * int(rng.integers(9, 56))
    s = int(rng.integers(1, 4))
    p = host.default_params()
    p.max_path_components = int(rng.choice([2, 3, 8, 128]))
    p.rr_threshold = float(rng.choice([1.0, 0.5, 0.0]))
    p.randomize_ray_over_pixel = int(rng.integers(0, 2))
    check("triangles seed %d" % seed, host.random_triangles(int(rng.integers(1, 3000)), seed, with_texcoords=bool(rng.integers(0, 2)), width=w, height=h,
                                                         aperture=float(rng.choice([0.0, 0.05]))), s, p)
    check("sponza-like seed %d" % seed, host.sponza_like(w, h, seed=seed % 1000 + 1, detail=0.03, tex_size=16, env_width=32,
                                                        importance_n=int(rng.choice([0, 8, 16]))), s, p, tables=True)
    check("courtyard-like seed %d" % seed, host.courtyard_like(w, h, seed=seed % 1000 + 1, triangles=int(rng.integers(2000, 20000)), tex_size=16), s, p)
    check("measured-like seed %d" % seed, host.measured_like(w, h, host.rgl_fixture("iso"), host.rgl_fixture("aniso"), seed=seed % 1000 + 1, detail=0.03,
                                                            tex_size=16, env_width=32, importance_n=8), s, p, tables=True)
    t0 = float(rng.random())
    t1 = t0 if rng.integers(0, 3) == 0 else t0 + float(rng.random()) * (1.2 - t0)
    p.t0, p.t1 = t0, t1
    check("animated variant %d [%g, %g]" % (r % 16, t0, t1), host.animated(w, h, int(r % 16), t0, t1), s, p)
    p.t0 = p.t1 = 0.0
    check("spheres variant %d" % (r % 5), host.spheres(w, h, r % 5), s, p, tables=(r % 5 == 4))
    sc = host.cornell(w, h, int(rng.integers(0, 2)), int(rng.choice([0, 2])))
    k = int(rng.integers(0, 4))
    if k:
        host.set_distortion(sc, k, k1=-0.2 * float(rng.random()), k2=0.05 * float(rng.random()), k3=-0.01 * float(rng.random()) if k != 1 else 0.0,
                            p1=0.001 * float(rng.random()) if k != 2 else 0.0, p2=-0.001 * float(rng.random()) if k != 2 else 0.0)
    host.set_camera_mode(sc, int(rng.integers(0, 3)), float(rng.choice([0.0, 0.065])))
    check("cornell lens %d" % k, sc, s, p)
    if (r + 1) % 5 == 0:
        print("%d scenes, %d mismatches, %.0f s" % (done, bad, time.time() - t_start), flush=True)
print("fuzz parity: %d scenes, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
