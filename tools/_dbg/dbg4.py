import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_parity import get_model
from torchani_amd import _lib
from bench import water_box
dev = torch.device('cuda:0')
variant = sys.argv[1]
sp_np, x_np, cell_np = water_box(18)
x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
sp = torch.from_numpy(sp_np).to(dev).to(torch.int32)
if variant == "where":
    base = torch.from_numpy(sp_np).to(dev)
    sp = torch.where(base == 0, 0, 3).to(torch.int32)
model = get_model("ani2x", 0, dev, neighborlist="cell")
eng = model.aev_computer.engine()
packed = model.neural_networks._pack(dev)
nbrs = eng.neighbors(sp, x, cell, (True, True, True), mode="cell", row_cap=160)
mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
aev = eng.forward(sp, nbrs, slab_mask=mask)
torch.cuda.synchronize()
print("species", sp.dtype, sp.shape, sp[0, :6].tolist(), "mask0", hex(int(mask[0]) & 0xffffffff), flush=True)
if variant == "flags0":
    packed.flags = 0
for it in range(3):
    ga = torch.zeros_like(aev)
    e, _, _ = packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask)
    torch.cuda.synchronize()
    print("call", it, "ok sum e", float(e.double().sum()), flush=True)
