"""Development: localise a force difference between the reference species numbering and compact_species."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402
from torchani_amd import _lib  # noqa: E402
from torchani_amd.engine import PackedNetworks  # noqa: E402
from torchani_amd.models import ANI2x  # noqa: E402

dev = torch.device("cuda:0")
sp_np, x_np, cell_np = water_box(30)
sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
pbc = (True, True, True)
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell", row_capacity=160)
model.compact_species = False
ref = model.energies_and_forces(sp, x, cell, pbc)
for name, flags in (("default", 0), ("no slab mask", _lib.MLP_FLAG_NO_SLAB_MASK), ("no fused", _lib.MLP_FLAG_NO_FUSED),
                    ("small tiles", _lib.MLP_FLAG_SMALL_TILES), ("d0 rows", _lib.MLP_FLAG_D0_ROWS)):
    PackedNetworks.default_flags = flags
    for compact in (False, True):
        model.compact_species = compact
        out = model.energies_and_forces(sp, x, cell, pbc)
        dF = (out.forces - ref.forces).abs()
        print(f"{name:14s} compact {compact!s:5s}: max|dE_atom| {float((out.atomic_energies - ref.atomic_energies).abs().max()):.2e}  "
              f"max|dF| {float(dF.max()):.2e}  atoms with |dF| > 1e-6: {int((dF.max(dim=-1).values > 1e-6).sum())}  "
              f"species of the worst atom {int(sp.view(-1)[int(dF.max(dim=-1).values.argmax())])}")
PackedNetworks.default_flags = 0
