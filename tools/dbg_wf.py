import sys, numpy as np
sys.path.insert(0, ".")
import torch
torch.zeros(1, device="cuda")
from wurblpt_amd import host, device as dev
from tests import oracle_loader
o = oracle_loader.load()
variant = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x40
w, h, s = 64, 48, int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc = host.cornell(w, h, 1, 2)
ref, _ = o.render(sc, s)
dev.lib().wpt_set_launch_config(0, variant)
got, _ = dev.DeviceScene(sc).render(s)
eq = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=2)
print("equal pixels", eq.sum(), "of", eq.size, "zero pixels", (got == 0).all(axis=2).sum(), "ref zero", (ref == 0).all(axis=2).sum())
print("nan", np.isnan(got).sum(), "rel", np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()))
bad = np.argwhere(~eq)
print(bad[:10])
for y, x in bad[:5]:
    print(y, x, got[y, x], ref[y, x])
