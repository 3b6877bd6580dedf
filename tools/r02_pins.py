"""Measures what the two reference-held statistical pins can be tightened to (GPU box):
  1. Cornell box (Lambertian, wurblpt-cornellbox.cpp) at 1024^2 x 4096 spp against the reference tree's converged
     Mitsuba render (tests/golden/cbox_mitsuba_64x64.npy: 64x64 block means of cbox-2500spp.exr);
  2. the reference's MIS test (wurblpt-mis-test.cpp): material sampling alone against MIS."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wurblpt_amd import device, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = np.load(os.path.join(ROOT, "tests", "golden", "cbox_mitsuba_64x64.npy"))
sc = host.cornell(1024, 1024)
ds = device.DeviceScene(sc)
for ssqrt in (8, 32, 64):
    t = time.time()
    img, _ = ds.render(ssqrt)
    dt = time.time() - t
    blocks = img[::-1].reshape(64, 16, 64, 16, 3).mean(axis=(1, 3))
    rel = np.sqrt(((blocks - ref) ** 2).sum() / (ref ** 2).sum())
    means = blocks.mean(axis=(0, 1)) / ref.mean(axis=(0, 1))
    b16 = blocks.reshape(16, 4, 16, 4, 3).mean(axis=(1, 3))
    r16 = ref.reshape(16, 4, 16, 4, 3).mean(axis=(1, 3))
    rel16 = np.sqrt(((b16 - r16) ** 2).sum() / (r16 ** 2).sum())
    print("cornell %4d spp %.1fs: 64x64 block rel-L2 %.5f, 16x16 block rel-L2 %.5f, channel mean ratios %s" % (ssqrt * ssqrt, dt, rel, rel16, means), flush=True)
del ds

W, H = 960, 540
imgs = {}
for hot in (False, True):
    sc = host.mis_test(W, H, hot)
    ds = device.DeviceScene(sc)
    for ssqrt in (10, 40):
        t = time.time()
        img, _ = ds.render(ssqrt)
        imgs[(hot, ssqrt)] = img
        print("mis hot=%d %d spp %.1fs mean %s max %.3g" % (hot, ssqrt * ssqrt, time.time() - t, img.mean(axis=(0, 1)), img.max()), flush=True)
    del ds
for ssqrt in (10, 40):
    a, b = imgs[(False, ssqrt)], imgs[(True, ssqrt)]
    for bs in (30, 60):
        ba = a.reshape(H // bs, bs, W // bs, bs, 3).mean(axis=(1, 3))
        bb = b.reshape(H // bs, bs, W // bs, bs, 3).mean(axis=(1, 3))
        rel = np.sqrt(((ba - bb) ** 2).sum() / (bb ** 2).sum())
        worst = np.abs(ba - bb).max() / bb.mean()
        print("mis %d spp, %dx%d blocks: rel-L2 %.5f, worst block |diff| / mean %.4f, mean ratio %.5f" % (ssqrt * ssqrt, bs, bs, rel, worst, a.mean() / b.mean()), flush=True)
# the two halves of the MIS image (same scene, different pixels' streams) as a noise yardstick: render twice at different size? use 1600 spp MIS vs 100 spp MIS
a, b = imgs[(True, 10)], imgs[(True, 40)]
ba = a.reshape(H // 30, 30, W // 30, 30, 3).mean(axis=(1, 3)); bb = b.reshape(H // 30, 30, W // 30, 30, 3).mean(axis=(1, 3))
print("mis 100 spp vs 1600 spp (both MIS), 30x30 blocks: rel-L2 %.5f" % np.sqrt(((ba - bb) ** 2).sum() / (bb ** 2).sum()))
