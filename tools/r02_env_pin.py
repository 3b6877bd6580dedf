"""Measures (GPU box) whether next-event estimation towards the importance-sampled environment map with MIS converges to
the same image as radiance on escape alone (no importance tables): the statistical pin of EnvironmentMap::p / d."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wurblpt_amd import device, host

W, H = 320, 180
imgs = {}
for n in (0, 32):
    sc = host.sponza_like(W, H, detail=0.25, tex_size=128, env_width=256, importance_n=n)
    ds = device.DeviceScene(sc)
    for s in (8, 32):
        t = time.time()
        imgs[(n, s)], _ = ds.render(s)
        print("importance_n %d, %d spp: %.1f s, mean %s" % (n, s * s, time.time() - t, imgs[(n, s)].mean(axis=(0, 1))), flush=True)
for s in (8, 32):
    a, b = imgs[(0, s)], imgs[(32, s)]
    for bs in (20, 45):
        ba = a.reshape(H // bs, bs, W // bs, bs, 3).mean(axis=(1, 3))
        bb = b.reshape(H // bs, bs, W // bs, bs, 3).mean(axis=(1, 3))
        print("%d spp, %dx%d blocks: rel-L2 %.5f, mean ratio %.5f, worst block %.4f" % (
            s * s, bs, bs, np.sqrt(((ba - bb) ** 2).sum() / (bb ** 2).sum()), a.mean() / b.mean(), np.abs(ba - bb).max() / bb.mean()), flush=True)
a, b = imgs[(32, 8)], imgs[(32, 32)]
ba = a.reshape(H // 20, 20, W // 20, 20, 3).mean(axis=(1, 3)); bb = b.reshape(H // 20, 20, W // 20, 20, 3).mean(axis=(1, 3))
print("with importance sampling, 64 vs 1024 spp, 20x20 blocks: rel-L2 %.5f" % np.sqrt(((ba - bb) ** 2).sum() / (bb ** 2).sum()))
a, b = imgs[(0, 8)], imgs[(0, 32)]
ba = a.reshape(H // 20, 20, W // 20, 20, 3).mean(axis=(1, 3)); bb = b.reshape(H // 20, 20, W // 20, 20, 3).mean(axis=(1, 3))
print("without, 64 vs 1024 spp, 20x20 blocks: rel-L2 %.5f" % np.sqrt(((ba - bb) ** 2).sum() / (bb ** 2).sum()))
