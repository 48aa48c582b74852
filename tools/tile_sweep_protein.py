"""Network stage of the solvated-1hz5 box (tests/golden/cfg3_*: five elements) replicated periodically, under the layer-0
tiling choices (development: up to which size 128-row tiles pay for many-element systems):  python tools/tile_sweep_protein.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import time_stage  # noqa: E402
from torchani_amd import _lib  # noqa: E402
from torchani_amd.engine import PackedNetworks  # noqa: E402
from torchani_amd.models import ANI2x  # noqa: E402

dev = torch.device("cuda:0")
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
eng = model.aev_computer.engine()
packed = model.neural_networks._pack(dev)
with np.load(os.path.join(ROOT, "tests", "golden", "cfg3_1hz5_water_ani2x.npz")) as z:
    sp0, x0, cell0 = z["species"].astype(np.int64)[0], z["coords"][0].astype(np.float32), z["cell"].astype(np.float32)
for rep in ((1, 1, 1), (2, 1, 1), (2, 2, 1), (2, 2, 2)):
    shifts = np.array([[i, j, k] for i in range(rep[0]) for j in range(rep[1]) for k in range(rep[2])], dtype=np.float32)
    x = (x0[None] + (shifts @ cell0)[:, None, :]).reshape(1, -1, 3)
    sp = np.tile(sp0, len(shifts))[None]
    cell = cell0 * np.asarray(rep, dtype=np.float32)[:, None]
    n = sp.shape[1]
    sp32 = torch.from_numpy(sp).to(dev).to(torch.int32).contiguous()
    coords, cl = torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
    nbrs = eng.neighbors(sp32, coords, cl, (True, True, True), mode="cell")
    mask = torch.zeros(n, dtype=torch.int32, device=dev)
    aev = eng.forward(sp32, nbrs, slab_mask=mask)
    ae = torch.zeros(n, dtype=torch.float32, device=dev)
    gaev = torch.zeros_like(aev)
    out = []
    for nm, fl in (("auto", 0), ("small", _lib.MLP_FLAG_SMALL_TILES), ("big", _lib.MLP_FLAG_BIG_TILES)):
        PackedNetworks.default_flags = fl
        t = min(time_stage(lambda: packed.forward_backward(sp32, aev, atomic_e=ae, grad_aev=gaev, slab_mask=mask), 5)
                for _ in range(2))
        out.append(f"{nm} {t:.3f} ms")
    PackedNetworks.default_flags = 0
    print(f"{n:8d} atoms: " + " | ".join(out), flush=True)
    del aev, gaev, nbrs
