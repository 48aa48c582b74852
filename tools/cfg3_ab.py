"""Config-3 inputs (tests/golden/cfg3_*: 1hz5 solvated 46 357 atoms periodic, 1C17 16 649 atoms) under anihip_mlp_desc.flags
variants, eager and graph-replayed (development A/B inside one gpurun call):  python tools/cfg3_ab.py 0 4"""
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
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
model.auto_graph_atoms = 0
for name in ("cfg3_1c17_ani2x", "cfg3_1hz5_water_ani2x"):
    with np.load(os.path.join(GOLD, name + ".npz")) as z:
        sp, x = z["species"].astype(np.int64), z["coords"]
        cell = z["cell"] if "cell" in z.files else None
    s3, c3 = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
    cl = None if cell is None else torch.from_numpy(cell).to(dev)
    pbc = None if cell is None else (True, True, True)
    res = []
    for fl in [int(a) for a in sys.argv[1:]] or [0]:
        PackedNetworks.default_flags = fl
        ms = min(timeit(lambda: model.energies_and_forces(s3, c3, cl, pbc, check_overflow=False), reps=20) for _ in range(3)) * 1e3
        g = model.graphed(s3, c3, cl, pbc)
        msg = min(timeit(lambda: g(c3), reps=20) for _ in range(3)) * 1e3
        res.append(f"flags={fl}: eager {ms:.3f} ms, graph {msg:.3f} ms")
    PackedNetworks.default_flags = 0
    print(name, sp.size, "atoms |", " | ".join(res))
