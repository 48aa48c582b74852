"""kernel timeline of ONE config-2 step (development aid): python tools/cfg2_trace.py  (under rocprofv3 --kernel-trace)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchani_amd.models import ANI2x
dev = torch.device("cuda:0")
with np.load(os.path.join(ROOT, "tests", "golden", "cfg2_xyz13_28_ani2x.npz")) as z:
    sp, x = z["species"].astype(np.int64), z["coords"]
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="batch")
spd, xd = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
if os.environ.get("AB_FLAGS"):   # (development: anihip_mlp_desc.flags variant, eager only)
    from torchani_amd.engine import PackedNetworks
    PackedNetworks.default_flags = int(os.environ["AB_FLAGS"])
    model.auto_graph_atoms = 0
if os.environ.get("AB_GRAPH"):   # the replayed HIP graph instead of the eager step
    model.auto_graph_atoms = 0
    f = model.graphed(spd, xd)
    for _ in range(20):
        f(xd)
else:
    for _ in range(3):
        model.energies_and_forces(spd, xd, check_overflow=False)
torch.cuda.synchronize()
