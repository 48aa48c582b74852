"""Half neighbor lists of the REFERENCE for some golden cases -> tests/golden/nbrs_<case>.npz.

    python tests/golden/gen_golden_nbrs.py         (needs /root/reference; the outputs are committed)

For every case the reference's own neighbor-list module (all_pairs, with PBC images where the case has a cell)
is run on the fixture's fp32 coordinates; the file stores Neighbors.indices [2, P] and Neighbors.diff_vectors
[P, 3] (fp32).  As a self-check the reference's AEVComputer.compute_from_neighbors on that list must reproduce
the AEV rows already stored in the fixture (they were computed through AEVComputer.forward).
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (bootstraps the reference import)

import torch  # noqa: E402
from torchani.aev import AEVComputer  # noqa: E402
from torchani.neighbors import AllPairs  # noqa: E402

CASES = ["simple2_ani2x", "rand_batch_ani2x", "water_pbc_ani2x", "triclinic_pbc_ani2x", "small_ani2x",
         "ch4_ani1x"]


def main():
    for name in CASES:
        g = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=True)
        kind = str(g["kind"])
        species = torch.from_numpy(g["species"].astype(np.int64))
        coords = torch.from_numpy(g["coords"]).double()
        cell = torch.from_numpy(g["cell"]).double() if "cell" in g.files else None
        pbc = torch.from_numpy(np.asarray(g["pbc"], dtype=bool)) if "pbc" in g.files else None
        aevc = (AEVComputer.like_2x() if kind == "ani2x" else AEVComputer.like_1x()).double()
        nl = AllPairs()
        nb = nl(aevc.radial.cutoff, species, coords, cell, pbc)
        aev = aevc.compute_from_neighbors(species, coords, nb)
        rows = g["aev_rows"]
        err = np.abs(aev.reshape(-1, aev.shape[-1]).numpy()[rows] - g["aev"]).max()
        assert err < 1e-12, (name, err)
        out = os.path.join(HERE, "nbrs_" + name + ".npz")
        np.savez_compressed(out, indices=nb.indices.numpy().astype(np.int64),
                            diff_vectors=nb.diff_vectors.numpy().astype(np.float32))
        print(f"{name}: {nb.indices.shape[1]} pairs, |aev(compute_from_neighbors) - fixture| = {err:.1e} -> {out}")


if __name__ == "__main__":
    main()
