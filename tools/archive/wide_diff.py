"""usage (GPU box): WPT_LIB_DIR=lib_wide python tools/wide_diff.py
Where the wide walk's prototype (-DWPT_WIDE_WALK) differs from the binary walk: the rendering kernel (wide) against the counting
kernel (binary, equal to the oracle) of the same library, pixel by pixel, for the scenes tools/fuzz_parity.py 6 3081 reported."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wurblpt_amd import device, host

for label, sc, s in (("spheres 4", host.spheres(48, 32, 4), 2),
                     ("sponza-like 333", host.sponza_like(48, 32, seed=333, detail=0.03, tex_size=16, env_width=32, importance_n=8), 2)):
    ds = device.DeviceScene(sc)
    p = host.default_params()
    for comps in (2, 3, 128):
        p.max_path_components = comps
        wide, _ = ds.render(s, params=p)
        binary, cnt = ds.render(s, params=p, with_counters=True)
        d = np.argwhere((wide.view(np.uint32) != binary.view(np.uint32)).any(axis=2))
        print(label, "components", comps, "pixels differing", len(d), "of", wide.shape[0] * wide.shape[1], "rays/sample %.2f" % (cnt["rays"] / cnt["samples"]), flush=True)
        for y, x in d[:6]:
            print("   pixel", int(x), int(y), "wide", wide[y, x].tolist(), "binary", binary[y, x].tolist(), flush=True)
