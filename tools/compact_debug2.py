"""Development: which columns of dE/dAEV differ between the skinny layer-0 backward and the row-major hand-over."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402
from torchani_amd import _lib  # noqa: E402
from torchani_amd.models import ANI2x  # noqa: E402

dev = torch.device("cuda:0")
sp_np, x_np, cell_np = water_box(30)
sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
pbc = (True, True, True)
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell", row_capacity=160)
eng = model.aev_computer.engine()
sp32 = sp.to(torch.int32)
for compact in (False, True):
    model.compact_species = compact
    sp_e, order = model._engine_species(sp32)
    packed = model.neural_networks._pack(dev, order)
    nbrs = eng.neighbors(sp_e, x, cell, pbc, mode="cell", row_cap=160)
    n = sp.numel()
    mask = torch.zeros(n, dtype=torch.int32, device=dev)
    aev = eng.forward(sp_e, nbrs, slab_mask=mask)
    out = {}
    for name, flags in (("l0b", 0), ("rows", _lib.MLP_FLAG_D0_ROWS)):
        packed.flags = flags
        ga = torch.zeros_like(aev)
        e, g, _ = packed.forward_backward(sp_e, aev, grad_aev=ga, slab_mask=mask)
        out[name] = (e.clone(), ga.clone())
    packed.flags = None
    d = (out["l0b"][1] - out["rows"][1]).abs()
    ref = out["rows"][1].abs()
    print(f"compact {compact}: order {order}  max|d grad_aev| {float(d.max()):.3e}  (max |grad_aev| {float(ref.max()):.3e})")
    blocks = [(f"radial sp{s}", 16 * s, 16 * s + 16) for s in range(7)] + [(f"ang P{p}", 112 + 32 * p, 144 + 32 * p) for p in range(28)]
    for nm, a, b in blocks:
        if float(ref[:, a:b].max()) > 0:
            for sname, sel in (("H", sp_e.view(-1) == 0), ("O", sp_e.view(-1) == (1 if compact else 3))):
                print(f"   {nm:12s} centres {sname}: max|d| {float(d[sel][:, a:b].max()):.3e}  max|ref| {float(ref[sel][:, a:b].max()):.3e}")
