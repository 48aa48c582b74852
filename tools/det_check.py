"""development check: which stage of energies_and_forces is not bit-reproducible?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box
from torchani_amd.models import ANI2x
dev = torch.device("cuda:0")
sp, x, cell = water_box(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
eng = model.aev_computer.engine()
sp32 = torch.from_numpy(sp).to(dev).to(torch.int32).contiguous()
xt, ct = torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
n = sp32.numel()
packed = model.neural_networks._pack(dev)
res = []
for rep in range(3):
    nbrs = eng.neighbors(sp32, xt, ct, (True, True, True), mode="cell")
    mask = torch.zeros(n, dtype=torch.int32, device=dev)
    aev = eng.forward(sp32, nbrs, slab_mask=mask)
    ga = torch.zeros_like(aev)
    ae, ga, _ = packed.forward_backward(sp32, aev, grad_aev=ga, slab_mask=mask)
    acc = eng.backward(sp32, nbrs, ga, slab_mask=mask, fixed_point=True)
    acc2 = eng.backward(sp32, nbrs, ga, slab_mask=mask, fixed_point=True)
    torch.cuda.synchronize()
    res.append((nbrs.ent.clone(), nbrs.meta.clone(), aev.clone(), ga.clone(), acc.clone()))
    print("rep", rep, "same-grad backward twice equal:", torch.equal(acc, acc2), int((acc - acc2).abs().max()))
for k, name in enumerate(["ent", "meta", "aev", "grad_aev", "acc"]):
    a, b = res[0][k], res[1][k]
    if a.dtype.is_floating_point:
        same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    else:
        same = torch.equal(a, b)
    print(name, "bit-equal across runs:", same, "" if same else float((a.double() - b.double()).abs().max()))
from torchani_amd import _lib
from torchani_amd.engine import PackedNetworks
for name, fl in (("default", 0), ("no_fused", _lib.MLP_FLAG_NO_FUSED), ("big_tiles", _lib.MLP_FLAG_BIG_TILES),
                 ("small_tiles", _lib.MLP_FLAG_SMALL_TILES), ("d0_rows", _lib.MLP_FLAG_D0_ROWS),
                 ("no_mask", _lib.MLP_FLAG_NO_SLAB_MASK)):
    PackedNetworks.default_flags = fl
    outs = []
    for rep in range(3):
        ga = torch.zeros_like(aev)
        ae, ga, _ = packed.forward_backward(sp32, aev, grad_aev=ga, slab_mask=mask)
        torch.cuda.synchronize()
        outs.append((ae.clone(), ga.clone()))
    print(f"{name:12s} e equal: {all(torch.equal(o[0], outs[0][0]) for o in outs)}  grad equal: "
          f"{all(torch.equal(o[1], outs[0][1]) for o in outs)}  max diff {max(float((o[1] - outs[0][1]).abs().max()) for o in outs):.3e}")
