"""Network stage (anihip_mlp_forward_backward) of periodic water boxes of growing size under the layer-0 tiling choices
(development: where the 256 x 256 tiling starts to pay):  python tools/tile_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_stage, water_box  # noqa: E402
from torchani_amd import _lib  # noqa: E402
from torchani_amd.engine import PackedNetworks  # noqa: E402
from torchani_amd.models import ANI2x  # noqa: E402

dev = torch.device("cuda:0")
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
eng = model.aev_computer.engine()
packed = model.neural_networks._pack(dev)
for side in [int(a) for a in sys.argv[1:]] or [18, 24, 30, 36, 44, 56, 92]:
    sp_np, x_np, cell_np = water_box(side)
    n = sp_np.shape[1]
    sp32 = torch.from_numpy(sp_np).to(dev).to(torch.int32).contiguous()
    coords, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    nbrs = eng.neighbors(sp32, coords, cell, (True, True, True), mode="cell")
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
