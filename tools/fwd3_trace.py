"""Per-phase shader-clock breakdown of k_aev_fwd3 (library built with -DANIHIP_TRACE; development)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402

NAMES = ["neighbor terms", "radial sums", "prefetch issue", "slot dealing", "iterator setup", "pair loop", "reduction",
         "wait prefetch", "stores", "loop head"]


def main():
    from torchani_amd import _lib
    from torchani_amd.models import ANI2x

    dev = torch.device("cuda:0")
    sp_np, x_np, cell_np = water_box(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
    sp32 = torch.from_numpy(sp_np).to(dev).to(torch.int32).contiguous()
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
    eng = model.aev_computer.engine()
    nbrs = eng.neighbors(sp32, torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev), (True, True, True), mode="cell")
    for _ in range(2):
        aev = eng.forward(sp32, nbrs)
    torch.cuda.synchronize()
    buf = np.zeros((2048, 10), dtype=np.uint64)
    rc = _lib.lib().anihip_dev_trace_read(buf.ctypes.data_as(C.c_void_p))
    assert rc == 0
    used = buf[buf.sum(axis=1) > 0].astype(np.float64)
    n_atoms = sp32.numel()
    per_wave_atoms = n_atoms / (used.shape[0] * 16)   # (mean: the waves of a workgroup share its atoms through a queue)
    tot = used.sum(axis=1).mean()
    print(f"blocks {used.shape[0]}  atoms per wave {per_wave_atoms:.1f}  clocks per atom (wave 0) {tot / per_wave_atoms:.0f}")
    tw = used.sum(axis=1)
    print(f"  per-wave totals (wave 0 of every block): min {tw.min() / tw.mean():.3f}  max {tw.max() / tw.mean():.3f} of the mean, "
          f"std {tw.std() / tw.mean():.4f}")
    for k, nm in enumerate(NAMES):
        print(f"  {nm:16s} {used[:, k].mean() / per_wave_atoms:8.0f} clocks/atom  {100 * used[:, k].mean() / tot:5.1f} %")


if __name__ == "__main__":
    main()
