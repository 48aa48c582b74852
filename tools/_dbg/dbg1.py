import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_parity import get_model
from torchani_amd import _lib
from bench import water_box
dev = torch.device('cuda:0')
sp_np, x_np, cell_np = water_box(18)
x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
base = torch.from_numpy(sp_np).to(dev)
for pair in [(0, 3), (1, 1)]:
    sp = torch.where(base == 0, pair[0], pair[1]).to(torch.int32)
    model = get_model("ani2x", 0, dev, neighborlist="cell")
    eng = model.aev_computer.engine()
    packed = model.neural_networks._pack(dev)
    nbrs = eng.neighbors(sp, x, cell, (True, True, True), mode="cell", row_cap=160)
    mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
    aev = eng.forward(sp, nbrs, slab_mask=mask)
    got = {}
    for name, flags in (("skinny", 0), ("skinny2", 0), ("rows", _lib.MLP_FLAG_D0_ROWS), ("rows2", _lib.MLP_FLAG_D0_ROWS)):
        packed.flags = flags
        ga = torch.zeros_like(aev)
        e, _, _ = packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask)
        got[name] = (e.clone(), ga.clone())
    packed.flags = None
    for a, b in (("skinny", "skinny2"), ("rows", "rows2"), ("skinny", "rows")):
        de = (got[a][0] - got[b][0]).abs()
        dg = (got[a][1] - got[b][1]).abs()
        print(pair, a, b, "dE max", float(de.max()), "n", int((de > 0).sum()), "dG max", float(dg.max()), "n", int((dg > 0).sum()), "Gmax", float(got[a][1].abs().max()))
