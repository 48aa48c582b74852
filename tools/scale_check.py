import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import water_box
from torchani_amd.models import ANI2x
dev = torch.device("cuda:0")
sp, x, cell = water_box(92)
spd = torch.from_numpy(sp.astype(np.int64)).to(dev); xd = torch.from_numpy(x).to(dev); cd = torch.from_numpy(cell).to(dev)
pbc = (True, True, True)
m = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="verlet_cell_list")
o1 = m.energies_and_forces(spd, xd, cd, pbc, stress=True, check_overflow=True)
x2 = xd + 0.1 * torch.randn_like(xd).clamp(-2, 2)
t0 = time.perf_counter(); o2 = m.energies_and_forces(spd, x2, cd, pbc, stress=True, check_overflow=True); torch.cuda.synchronize(); t1 = time.perf_counter()
m2 = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell_list")
o3 = m2.energies_and_forces(spd, x2, cd, pbc, stress=True, check_overflow=True)
v = m.aev_computer.verlet
print("atoms", sp.size, "verlet builds/reuses", v.n_builds, v.n_reuses, "step ms", (t1 - t0) * 1e3)
print("E diff", abs(o2.energies.item() - o3.energies.item()), "F diff", (o2.forces - o3.forces).abs().max().item(),
      "virial diff", (o2.virial - o3.virial).abs().max().item(), "virial trace/3V (Ha/A^3)", (o3.virial.trace() / 3 / torch.det(cd.double())).item())
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)
