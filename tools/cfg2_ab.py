"""Graph-replayed config-2 step (tests/golden/cfg2_*: 256 molecules, 5248 real atoms) under anihip_mlp_desc.flags
variants (development A/B inside one gpurun call):  [TORCHANI_AMD_LIB=alt.so] python tools/cfg2_ab.py 0 256 512"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_configs import GOLD, timeit  # noqa: E402
from torchani_amd.engine import PackedNetworks  # noqa: E402
from torchani_amd.models import ANI2x  # noqa: E402

dev = torch.device("cuda:0")
with np.load(os.path.join(GOLD, "cfg2_xyz13_28_ani2x.npz")) as z:
    sp, x = z["species"].astype(np.int64), z["coords"]
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="batch")
spd, xd = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
model.auto_graph_atoms = 0
res = []
for fl in [int(a) for a in sys.argv[1:]] or [0]:
    PackedNetworks.default_flags = fl
    g = model.graphed(spd, xd)
    ms = min(timeit(lambda: g(xd), reps=100) for _ in range(3)) * 1e3
    res.append(f"flags={fl}: {ms:.4f} ms")
print(os.environ.get("TORCHANI_AMD_LIB", "default lib"), " | ".join(res))
