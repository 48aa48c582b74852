import sys, ctypes, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_parity import get_model
from torchani_amd import _lib
from bench import water_box
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
P = ctypes.CDLL('tools/_dbg/libpoison.so')
sp_np, x_np, cell_np = water_box(18)
x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
base = torch.from_numpy(sp_np).to(dev)
rs = np.random.RandomState(1)
for elements in [(1, 1), (0, 3), (0, 1, 2, 3, 4, 5, 6)]:
    spn = rs.choice(np.asarray(elements), size=sp_np.shape).astype(np.int32)
    sp = torch.from_numpy(spn).to(dev)
    model = get_model("ani2x", 0, dev, neighborlist="cell")
    eng = model.aev_computer.engine()
    packed = model.neural_networks._pack(dev)
    nbrs = eng.neighbors(sp, x, cell, (True, True, True), mode="cell", row_cap=192)
    mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
    aev = eng.forward(sp, nbrs, slab_mask=mask)
    ref = None
    for name, flags, pat in (("plain", 0, None), ("plain", 0, None), ("plain+poisonNaN", 0, 0x7E007E00), ("plain+poison big", 0, 0x7BFF7BFF), ("plain+poison0", 0, 0),
                             ("l0b", _lib.MLP_FLAG_FUSED_L0B, None), ("l0b+poisonNaN", _lib.MLP_FLAG_FUSED_L0B, 0x7E007E00), ("l0b+poison0", _lib.MLP_FLAG_FUSED_L0B, 0)):
        if pat is not None:
            P.poison_lds(ctypes.c_uint(pat))
        packed.flags = flags
        ga = torch.zeros_like(aev)
        e, _, _ = packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask)
        torch.cuda.synchronize()
        if ref is None:
            ref = (e.clone(), ga.clone())
        de = (e - ref[0]).abs(); dg = (ga - ref[1]).abs()
        print(elements, name, "dE max %.3e n %d  dG max %.3e n %d  nan %d" % (float(de.nan_to_num(1e9).max()), int((de > 0).sum()), float(dg.nan_to_num(1e9).max()), int((dg > 0).sum()), int(torch.isnan(e).sum()) + int(torch.isnan(ga).sum())))
    packed.flags = None
