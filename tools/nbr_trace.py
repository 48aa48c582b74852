"""Per-phase shader-clock breakdown of k_nbr_cell2 (library built with -DANIHIP_TRACE; development)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402

NAMES = ["loop tail", "ownership, stencil, prefix sum", "staging + class mask", "sweeps", "rows"]


def main():
    from torchani_amd import _lib
    from torchani_amd.models import ANI2x

    dev = torch.device("cuda:0")
    sp_np, x_np, cell_np = water_box(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
    sp32 = torch.from_numpy(sp_np).to(dev).to(torch.int32).contiguous()
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
    eng = model.aev_computer.engine()
    x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    for _ in range(3):
        eng.neighbors(sp32, x, cell, (True, True, True), mode="cell", row_cap=128)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        eng.neighbors(sp32, x, cell, (True, True, True), mode="cell", row_cap=128)
    t1.record(); torch.cuda.synchronize()
    n = sp32.numel()
    print(f"neighbors stage: {t0.elapsed_time(t1) / 5:.3f} ms for {n} atoms")
    buf = np.zeros((1024, 10), dtype=np.uint64)
    fn = getattr(_lib.lib(), "anihip_dev_nbr_trace_read", None)
    if fn is None:
        return
    assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
    used = buf[buf[:, :5].sum(axis=1) > 0].astype(np.float64)
    bins = used[:, 8].mean()
    tot = used[:, :5].sum(axis=1).mean()
    print(f"k_nbr_cell2: blocks {used.shape[0]}  bins per wave {bins:.1f}  clocks per bin (wave 0) {tot / bins:.0f}")
    for k, nm in enumerate(NAMES):
        print(f"  {nm:32s} {used[:, k].mean() / bins:8.0f} clocks/bin  {100 * used[:, k].mean() / tot:5.1f} %")


if __name__ == "__main__":
    main()
