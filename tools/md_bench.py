"""MD step rate with and without Verlet-skin neighbor reuse (cf. the reference's tools/md-benchmark.py, which drives
the same path through ASE):   python tools/md_bench.py [--side 64] [--steps 20]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import water_box  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--force-verlet", action="store_true",
                    help="refresh the skin list at every size (VerletRows.rebuild_above = inf): how the crossover was measured")
    args = ap.parse_args()
    from torchani_amd.md import MolecularDynamics
    from torchani_amd.models import ANI2x

    dev = torch.device("cuda:0")
    sp, x, cell = water_box(args.side)
    spd = torch.from_numpy(sp.astype(np.int64)).to(dev)
    xd, cd = torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
    masses = torch.tensor([1.008, 12.011, 14.007, 15.999, 32.06, 18.998, 35.45], device=dev)[spd]
    for nl in ("cell_list", "verlet_cell_list"):
        model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist=nl)
        if args.force_verlet and model.aev_computer.verlet is not None:
            model.aev_computer.verlet.rebuild_above = float("inf")
        md = MolecularDynamics(model, spd, xd, cd, (True, True, True), dt=0.5, masses=masses, seed=1)
        md.set_temperature(300.0)
        md.run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        md.run(args.steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ver = model.aev_computer.verlet
        extra = (f", pair searches {ver.n_builds}, reuses {ver.n_reuses}, steps rebuilt outright {ver.n_direct}"
                 if ver is not None else "")
        print(f"{nl:17s} {sp.size} atoms: {dt * 1e3:.2f} ms/step = {sp.size / dt / 1e6:.2f} M atom*steps/s, "
              f"T = {md.temperatures().item():.0f} K{extra}")


if __name__ == "__main__":
    main()
