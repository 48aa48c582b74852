"""Development: skinny layer-0 backward vs row-major hand-over for two-species systems of every pairing (reference numbering)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402
from torchani_amd import _lib  # noqa: E402
from torchani_amd.models import ANI2x  # noqa: E402

dev = torch.device("cuda:0")
sp_np, x_np, cell_np = water_box(30)
x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
pbc = (True, True, True)
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell", row_capacity=160)
model.compact_species = False
eng = model.aev_computer.engine()
packed = model.neural_networks._pack(dev)
base = torch.from_numpy(sp_np).to(dev)
for a, b in ((0, 3), (0, 1), (0, 2), (1, 2), (2, 3), (4, 5), (0, 6)):
    sp = torch.where(base == 0, a, b).to(torch.int32)
    nbrs = eng.neighbors(sp, x, cell, pbc, mode="cell", row_cap=160)
    n = sp.numel()
    mask = torch.zeros(n, dtype=torch.int32, device=dev)
    aev = eng.forward(sp, nbrs, slab_mask=mask)
    out = {}
    for name, flags in (("l0b", 0), ("rows", _lib.MLP_FLAG_D0_ROWS)):
        packed.flags = flags
        ga = torch.zeros_like(aev)
        packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask)
        out[name] = ga.clone()
    packed.flags = None
    pop = int(mask[0].item()) & 0xFFFFFFFF
    d = (out["l0b"] - out["rows"]).abs()
    cols = torch.nonzero(d.max(dim=0).values > 0).view(-1)
    print(f"species ({a},{b}): mask of atom 0 {pop:#x} ({bin(pop).count('1')} slabs)  max|d| {float(d.max()):.3e}  "
          f"differing columns {cols.min().item() if cols.numel() else '-'}..{cols.max().item() if cols.numel() else '-'} ({cols.numel()})")
