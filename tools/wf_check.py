"""usage (GPU box): python tools/wf_check.py [parity|time] ...
parity [rounds [seed]]: seeded random scenes of every family that has a wavefront form, rendered by the wavefront kernels
    (wpt_set_wavefront mode 1; by kind of material and in ray queue order; 1 to 3 groups; whole frame, a block, bands)
    against the single kernel: frames must agree bit for bit.
time workload samples_sqrt [groups chunk flags]...: frame time of the workload with the single kernel and with the
    wavefront kernels in the given configurations."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wurblpt_amd import device, host

L = device.lib()


def render(ds, s, p, mode, groups=0, chunk=0, flags=0, block=None):
    L.wpt_set_wavefront(mode, groups, chunk, flags)
    try:
        got, _ = ds.render(s, params=p, block=block)
        ds.check()
        return got
    finally:
        L.wpt_set_wavefront(0, 0, 0, 0)


def parity(rounds, seed):
    rng = np.random.default_rng(seed)
    bad = done = 0
    t_start = time.time()

    def check(label, sc, s, p):
        nonlocal bad, done
        ds = device.DeviceScene(sc)
        ref = render(ds, s, p, 2)
        h_, w_ = ref.shape[:2]
        for groups, chunk, flags in ((1, 0, 0), (2, 64, 1 | (7 << 16)), (3, 32, 0x800 | (1 << 16)), (2, 0, 0x2000 | (0xffff << 16)), (2, 0, 33 << 16)):
            got = render(ds, s, p, 1, groups, chunk, flags)
            n = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
            if n:
                bad += 1
                print("MISMATCH %s groups %d chunk %d flags %#x: %d values differ" % (label, groups, chunk, flags, n), flush=True)
        start = int(rng.integers(0, w_ * h_))
        size = int(rng.integers(1, w_ * h_ - start + 1))
        part = render(ds, s, p, 1, 2, 0, 0, block=(start, size))
        n = int((part.reshape(-1, 3)[start:start + size].view(np.uint32) != ref.reshape(-1, 3)[start:start + size].view(np.uint32)).sum())
        n += int(np.count_nonzero(part.reshape(-1, 3)[:start])) + int(np.count_nonzero(part.reshape(-1, 3)[start + size:]))
        band_rows, stride = int(rng.integers(1, 20)), int(rng.integers(1, 4))
        total = np.zeros_like(ref)
        L.wpt_set_wavefront(1, 0, 0, 0)
        for rank in range(stride):
            fr = torch.zeros((h_, w_, 3), dtype=torch.float32, device="cuda")
            ds.render_bands_into(fr, s, band_rows, rank, stride, params=p, stream=torch.cuda.current_stream())
            torch.cuda.synchronize()
            total += fr.cpu().numpy()
        L.wpt_set_wavefront(0, 0, 0, 0)
        n += int((total.view(np.uint32) != ref.view(np.uint32)).sum())
        if n:
            bad += 1
            print("MISMATCH %s block / bands: %d values differ" % (label, n), flush=True)
        done += 1

    for r in range(rounds):
        sd = int(rng.integers(1, 1 << 30))
        w, h = 8 * int(rng.integers(6, 40)), 8 * int(rng.integers(5, 30))
        s = int(rng.integers(1, 6))
        p = host.default_params()
        p.max_path_components = int(rng.choice([3, 8, 128]))
        p.rr_threshold = float(rng.choice([1.0, 0.5]))
        check("triangles seed %d" % sd, host.random_triangles(int(rng.integers(100, 3000)), sd, with_texcoords=bool(rng.integers(0, 2)), width=w, height=h,
                                                           aperture=float(rng.choice([0.0, 0.05]))), s, p)
        check("sponza-like seed %d" % sd, host.sponza_like(w, h, seed=sd % 1000 + 1, detail=0.03, tex_size=16, env_width=32,
                                                          importance_n=int(rng.choice([0, 8, 16]))), s, p)
        check("courtyard-like seed %d" % sd, host.courtyard_like(w, h, seed=sd % 1000 + 1, triangles=int(rng.integers(2000, 20000)), tex_size=16), s, p)
        check("measured-like seed %d" % sd, host.measured_like(w, h, host.rgl_fixture("iso"), host.rgl_fixture("aniso"), seed=sd % 1000 + 1, detail=0.03,
                                                              tex_size=16, env_width=32, importance_n=8), s, p)
        check("spheres variant %d" % (r % 5), host.spheres(w, h, r % 5), s, p)
        check("cornell", host.cornell(w, h, int(rng.integers(0, 2)), int(rng.choice([0, 2]))), s, p)
        print("%d scenes, %d mismatches, %.0f s" % (done, bad, time.time() - t_start), flush=True)
    print("wavefront parity: %d scenes, %d mismatches" % (done, bad))
    return 1 if bad else 0


def timing(workload, s, configs):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    w = dict(bench.WORKLOADS[workload])
    t0 = time.time()
    sc = bench.build_scene(w)
    ds = device.DeviceScene(sc)
    print("scene %s built and uploaded in %.1f s" % (workload, time.time() - t0), flush=True)
    width, height = w["width"], w["height"]
    p = host.default_params()
    frame = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()

    def run(label, mode, groups=0, chunk=0, flags=0, reps=int(os.environ.get('WF_REPS', '2'))):
        L.wpt_set_wavefront(mode, groups, chunk, flags)
        best = None
        for _ in range(reps):
            frame.zero_()
            torch.cuda.synchronize()
            t = time.perf_counter()
            ds.render_block_into(frame, s, None, p, None, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        L.wpt_set_wavefront(0, 0, 0, 0)
        out = frame.cpu().numpy().copy()
        print("%-40s %8.1f ms  %7.1f Msamples/s  launches %d" % (label, best * 1e3, width * height * s * s / best / 1e6, L.wpt_last_render_passes()), flush=True)
        return out

    ref = run("single kernel", 2)
    for groups, chunk, flags in configs:
        got = run("wavefront groups %d chunk %d flags %#x" % (groups, chunk, flags), 1, groups, chunk, flags)
        n = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
        if n:
            print("   MISMATCH: %d values differ" % n, flush=True)
    return 0


if __name__ == "__main__":
    if sys.argv[1] == "parity":
        sys.exit(parity(int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 7))
    configs = [tuple(int(x, 0) for x in c.split(",")) for c in sys.argv[4:]] or [(2, 0, 0)]
    sys.exit(timing(sys.argv[2], int(sys.argv[3]), configs))
