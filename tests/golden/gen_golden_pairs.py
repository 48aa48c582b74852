"""Golden fixtures of the xTB repulsion pair potential, from the reference's RepulsionXTB in float64
(torchani/potentials/xtb.py; envelope / halves of potentials/core.py) on the coordinates of existing golden cases:

    python tests/golden/gen_golden_pairs.py      -> tests/golden/pairs_<case>.npz

Each file: per-atom energies (atomic=True), molecular energies and forces for cutoff 5.2 A with the smooth envelope
(the ANI-2xr recipe, arch.py:1055-1060), and for water_pbc also with cutoff 5.1 / cosine.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the reference import)

import torch  # noqa: E402
from torchani.potentials import RepulsionXTB  # noqa: E402

ZNUM = {"H": 1, "C": 6, "N": 7, "O": 8, "S": 16, "F": 9, "Cl": 17}


def run(name, cutoff, cutoff_fn, tag):
    with np.load(os.path.join(HERE, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = [str(s) for s in g["symbols"]]
    pot = RepulsionXTB(symbols=symbols, cutoff=cutoff, cutoff_fn=cutoff_fn).double()
    elem = torch.from_numpy(g["species"].astype(np.int64))
    coords = torch.from_numpy(g["coords"]).double().requires_grad_(True)
    cell = torch.from_numpy(g["cell"]).double() if "cell" in g else None
    pbc = torch.from_numpy(g["pbc"]) if "pbc" in g else None
    # (Potential.forward passes ``atomic`` into the ``charge`` slot, core.py:65-67: call compute_from_neighbors itself)
    from torchani.neighbors import all_pairs

    neighbors = all_pairs(cutoff, elem, coords, cell, pbc)
    atomic = pot.compute_from_neighbors(elem, coords, neighbors, atomic=True).energies
    e = atomic.sum(dim=1)
    (grad,) = torch.autograd.grad(e.sum(), coords)
    out = dict(cutoff=np.asarray(cutoff), cutoff_fn=np.asarray(cutoff_fn), atomic_energies=atomic.detach().numpy(),
               energies=e.detach().numpy(), forces=(-grad).numpy())
    path = os.path.join(HERE, f"pairs_{tag}{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: E[0]={e[0].item():+.9f} |F|max={grad.abs().max().item():.5f}")


if __name__ == "__main__":
    for nm in ("rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x", "triclinic_pbc_ani2x"):
        run(nm, 5.2, "smooth", "")
    run("water_pbc_ani2x", 5.1, "cosine", "cos_")
