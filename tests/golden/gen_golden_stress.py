"""Golden virials: the reference's "scaling" stress (ase.py:115-173: coords and cell multiplied by a strain matrix,
stress = dE/d(scaling) / volume) evaluated with autograd in fp64 for periodic base fixtures.

    python tests/golden/gen_golden_stress.py      (needs /root/reference; the outputs are committed)

Stored: virial = dE_nn / d(scaling) [3,3] in Hartree (NN energy only; the self energies do not depend on the strain).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def run_case(base):
    torch = gg.torch
    with np.load(os.path.join(gg.HERE, base + ".npz")) as z:
        b = {k: z[k] for k in z.files}
    kind, seed = str(b["kind"]), int(b["seed"])
    gg.CUTOFF_FN = str(b["cutoff_fn"]) if "cutoff_fn" in b else "cosine"
    model = gg.build_reference(kind, seed)
    model.set_enabled("energy_shifter", False)
    elem = torch.as_tensor(b["species"].astype(np.int64))
    coords = torch.as_tensor(b["coords"]).double()
    cell = torch.as_tensor(b["cell"]).double()
    pbc = torch.as_tensor(b["pbc"])
    scaling = torch.eye(3, dtype=torch.float64, requires_grad=True)
    znums = model.species_converter.atomic_numbers[elem.clamp(min=0)].masked_fill(elem < 0, -1) \
        if hasattr(model.species_converter, "atomic_numbers") else None
    aev = model.aev_computer(elem, coords @ scaling, cell @ scaling, pbc)
    nets = model.potentials["nnp"].neural_networks if hasattr(model, "potentials") else model.neural_networks
    e = nets(elem, aev).sum()
    (vir,) = torch.autograd.grad(e, scaling)
    volume = float(torch.det(cell).abs())
    path = os.path.join(gg.HERE, "stress_" + base + ".npz")
    np.savez_compressed(path, base=np.asarray(base), virial=vir.numpy(), volume=np.asarray(volume),
                        energy_nn=np.asarray(float(e)))
    print(f"stress_{base}: E_nn={float(e):+.9f} volume={volume:.3f}\n{vir.numpy()}")


def main():
    gg.torch.set_num_threads(8)
    for base in ("water_pbc_ani2x", "triclinic_pbc_ani2x", "benzene_pbc_ani2x", "water_pbc_smooth_ani2x"):
        run_case(base)


if __name__ == "__main__":
    main()
