import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_parity import get_model
from torchani_amd import _lib
from bench import water_box
dev = torch.device('cuda:0')
side = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sp_np, x_np, cell_np = water_box(side)
x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
sp = torch.from_numpy(sp_np).to(dev).to(torch.int32)
model = get_model("ani2x", 0, dev, neighborlist="cell")
eng = model.aev_computer.engine()
packed = model.neural_networks._pack(dev)
print("nbr", flush=True)
nbrs = eng.neighbors(sp, x, cell, (True, True, True), mode="cell", row_cap=160)
torch.cuda.synchronize(); print("aev", flush=True)
mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
aev = eng.forward(sp, nbrs, slab_mask=mask)
torch.cuda.synchronize(); print("mlp", flush=True)
ref = None
for it in range(12):
    ga = torch.zeros_like(aev)
    e, me, _ = packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask, member_energies=True) if False else packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask)
    torch.cuda.synchronize()
    if ref is None:
        ref = (e.clone(), ga.clone()); continue
    de = (e - ref[0]).abs(); dg = (ga - ref[1]).abs()
    idx = torch.nonzero(de > 0).flatten()
    print("run", it, "atoms", sp.numel(), "dE max %.3e n %d  dG max %.3e n %d" % (float(de.max()), int((de > 0).sum()), float(dg.max()), int((dg > 0).sum())),
          "first idx", idx[:12].tolist(), "species", sp.flatten()[idx[:12]].tolist(), "rel", (de[idx[:6]] / ref[0][idx[:6]].abs()).tolist())
