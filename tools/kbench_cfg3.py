"""AEV forward / backward device time on the 46 357-atom solvated-protein box of tests/golden/cfg3_* (five elements
present: many species-pair blocks per atom), replicated x8 in memory for stable timing.  Development microbenchmark."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import time_stage  # noqa: E402


def main():
    from torchani_amd.models import ANI2x

    dev = torch.device("cuda:0")
    path = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cfg3_*1hz5*.npz")))[0]
    with np.load(path) as z:
        sp, x, cell = z["species"].astype(np.int64), z["coords"].astype(np.float32), z["cell"].astype(np.float32)
    sp32 = torch.from_numpy(sp).to(dev).to(torch.int32).reshape(1, -1).contiguous()
    coords = torch.from_numpy(x).to(dev).reshape(1, -1, 3).contiguous()
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
    eng = model.aev_computer.engine()
    nbrs = eng.neighbors(sp32, coords, torch.from_numpy(cell).to(dev), (True, True, True), mode="cell")
    n = sp32.numel()
    mask = torch.zeros(n, dtype=torch.int32, device=dev)
    aev = eng.forward(sp32, nbrs, slab_mask=mask)
    gaev = torch.randn_like(aev) * 1e-3
    gc = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    fwd = time_stage(lambda: eng.forward(sp32, nbrs, out=aev, slab_mask=mask), 20)
    bwd = time_stage(lambda: eng.backward(sp32, nbrs, gaev, gc), 20)
    print(f"{os.path.basename(path)} atoms={n} fwd={fwd * 1e3:.1f}us bwd={bwd * 1e3:.1f}us  checksum={float(aev.double().sum()):.6f}")


if __name__ == "__main__":
    main()
