"""Throughput of the secondary BASELINE.json configurations on one GPU (development benchmark; the headline
metric is bench.py):   python tools/bench_configs.py
  config 2: 256 molecules x ~20 atoms (synthetic H/C/N/O, SURVEY 8d fallback input), batch mode
  config 3: 46 875-atom periodic water box, cell mode
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402


def molecules(n_mol=256, n_at=20, seed=2):
    rs = np.random.RandomState(seed)
    sp = rs.choice([0, 1, 2, 3], size=(n_mol, n_at), p=[0.5, 0.3, 0.1, 0.1])
    x = np.zeros((n_mol, n_at, 3), dtype=np.float32)
    for m in range(n_mol):
        pts = []
        while len(pts) < n_at:
            p = rs.uniform(0, 6.0, 3)
            if all(np.linalg.norm(p - q) > 0.9 for q in pts):
                pts.append(p)
        x[m] = np.asarray(pts, dtype=np.float32)
    return sp.astype(np.int64), x


def timeit(fn, warm=5, reps=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    from torchani_amd.models import ANI2x

    dev = torch.device("cuda:0")
    sp, x = molecules()
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="batch")
    spd, xd = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
    dt = timeit(lambda: model.energies_and_forces(spd, xd, check_overflow=False))
    print(f"config 2: 256 x 20 atoms, batch mode: {dt * 1e3:.3f} ms/step, {sp.size / dt / 1e6:.2f} M atom*steps/s")
    f = model.graphed(spd, xd)
    dt = timeit(lambda: f(xd))
    print(f"config 2, HIP graph replay:           {dt * 1e3:.3f} ms/step, {sp.size / dt / 1e6:.2f} M atom*steps/s")
    sp3, x3, cell = water_box(25)
    model3 = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
    s3, c3, cl = torch.from_numpy(sp3).to(dev), torch.from_numpy(x3).to(dev), torch.from_numpy(cell).to(dev)
    dt = timeit(lambda: model3.energies_and_forces(s3, c3, cl, (True, True, True), check_overflow=False), reps=20)
    print(f"config 3: {sp3.size}-atom periodic water box: {dt * 1e3:.3f} ms/step, {sp3.size / dt / 1e6:.2f} M atom*steps/s")


if __name__ == "__main__":
    main()
