"""Golden fixtures of the other closed-form pair potentials of torchani.potentials, from the reference's own classes in
float64 (zbl.py, lj.py, fixed_coulomb.py; envelope / halves of core.py) on the coordinates of existing golden cases:

    python tests/golden/gen_golden_pairs2.py      -> tests/golden/pairs2_<case>.npz

Per potential: per-atom energies (atomic=True) and forces; constructor arguments are stored with the values.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the reference import)

import torch  # noqa: E402
from torchani.neighbors import all_pairs  # noqa: E402
from torchani.potentials import (DispersionLJ, FixedCoulomb, FixedMNOK, LennardJones, RepulsionLJ,  # noqa: E402
                                 RepulsionZBL)

CHARGES = {"H": 0.35, "C": -0.15, "N": -0.45, "O": -0.70, "S": -0.20, "F": -0.25, "Cl": -0.10}
ETA = {"H": 0.47, "C": 0.37, "N": 0.53, "O": 0.45, "S": 0.30, "F": 0.52, "Cl": 0.34}
EPS = {"H": 2.5e-5, "C": 1.4e-4, "N": 2.7e-4, "O": 3.3e-4, "S": 4.0e-4, "F": 1.0e-4, "Cl": 4.2e-4}
SIGMA = {"H": 1.49, "C": 1.91, "N": 1.82, "O": 1.66, "S": 1.98, "F": 1.75, "Cl": 1.95}


def cases(symbols):
    q = tuple(CHARGES[s] for s in symbols)
    return {
        "zbl": (RepulsionZBL, dict(cutoff=5.2, cutoff_fn="smooth")),
        "zbl_cos": (RepulsionZBL, dict(k=0.4685, cutoff=4.0, cutoff_fn="cosine")),
        "lj": (LennardJones, dict(eps=tuple(EPS[s] for s in symbols), sigma=tuple(SIGMA[s] for s in symbols), cutoff=7.5,
                                  cutoff_fn="smooth")),
        "lj_rep": (RepulsionLJ, dict(cutoff=5.2, cutoff_fn="smooth")),
        "lj_disp": (DispersionLJ, dict(cutoff=7.5, cutoff_fn="smooth")),
        "coulomb": (FixedCoulomb, dict(charges=q, dielectric=1.3, cutoff=7.5, cutoff_fn="smooth")),
        "mnok": (FixedMNOK, dict(charges=q, eta=tuple(ETA[s] for s in symbols), cutoff=7.5, cutoff_fn="smooth")),
    }


def run(name):
    with np.load(os.path.join(HERE, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = [str(s) for s in g["symbols"]]
    elem = torch.from_numpy(g["species"].astype(np.int64))
    cell = torch.from_numpy(g["cell"]).double() if "cell" in g else None
    pbc = torch.from_numpy(g["pbc"]) if "pbc" in g else None
    out = {}
    for key, (cls, kw) in cases(symbols).items():
        pot = cls(symbols=symbols, **kw).double()
        coords = torch.from_numpy(g["coords"]).double().requires_grad_(True)
        neighbors = all_pairs(kw["cutoff"], elem, coords, cell, pbc)
        atomic = pot.compute_from_neighbors(elem, coords, neighbors, atomic=True).energies
        (grad,) = torch.autograd.grad(atomic.sum(), coords)
        out[key + "_atomic"] = atomic.detach().numpy()
        out[key + "_forces"] = (-grad).numpy()
        print(f"{name} {key:8s} E[0]={atomic.sum(dim=1)[0].item():+.9f} |F|max={grad.abs().max().item():.5f}")
    np.savez_compressed(os.path.join(HERE, f"pairs2_{name}.npz"), **out)


if __name__ == "__main__":
    for nm in ("rand_batch_ani2x", "water_pbc_ani2x", "triclinic_pbc_ani2x"):
        run(nm)
