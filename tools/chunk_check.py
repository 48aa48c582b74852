import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import time_stage, water_box
from torchani_amd.models import ANI2x
dev = torch.device("cuda:0")
sp_np, x_np, cell_np = water_box(92)
n = sp_np.shape[1]
sp32 = torch.from_numpy(sp_np).to(dev).to(torch.int32).contiguous()
coords = torch.from_numpy(x_np).to(dev); cell = torch.from_numpy(cell_np).to(dev)
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
eng = model.aev_computer.engine(); packed = model.neural_networks._pack(dev)
nbrs = eng.neighbors(sp32, coords, cell, (True, True, True), mode="cell")
mask = torch.zeros(n, dtype=torch.int32, device=dev)
aev = eng.forward(sp32, nbrs, slab_mask=mask)
ae = torch.zeros(n, dtype=torch.float32, device=dev); gaev = torch.zeros_like(aev)
for ch in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 22):
    t = time_stage(lambda: packed.forward_backward(sp32, aev, atomic_e=ae, grad_aev=gaev, slab_mask=mask, chunk=ch), 5)
    print(f"chunk {ch:8d}: mlp {t:.3f} ms", flush=True)
