"""Golden fixtures for torchani_amd.models.simple_ani from the reference's own flexible builder (arch.py:992-1066), fp64, with
torchani_amd.weights.random_network_state_dict as parameters (simple_ani models start from random parameters in both):

    python tests/golden/gen_golden_simple.py     -> tests/golden/simple_<case>.npz

case "chno":  H C N O, two members, default (ANI-2x widths, GELU, no biases, smooth envelope, xTB repulsion)
case "chno1x": H C N O, one member, like_1x widths, CELU with biases, cosine cutoff, 4 x 8 angular grid, no repulsion
Inputs: the first 12 molecules of the config-2 batch (tests/golden/cfg2_xyz13_28_ani2x.npz).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_d3 as g3  # noqa: E402,F401  (reference import + h5py stand-in)

import torch  # noqa: E402
from torchani.arch import simple_ani  # noqa: E402

from torchani_amd.constants import HIDDEN_DIMS_1X, HIDDEN_DIMS_2X  # noqa: E402
from torchani_amd.weights import random_network_state_dict  # noqa: E402

SYM = ("H", "C", "N", "O")
CASES = {
    "chno": dict(kw=dict(ensemble_size=2), hidden=HIDDEN_DIMS_2X, bias=False, seed=41),
    "chno1x": dict(kw=dict(ensemble_size=1, container_ctor="like_1x", activation="celu", bias=True, cutoff_fn="cosine",
                           angular_shifts=4, sections=8, angular_precision=8.0, angular_zeta=32.0, radial_precision=16.0,
                           repulsion=False), hidden=HIDDEN_DIMS_1X, bias=True, seed=42),
}


def run(case):
    c = CASES[case]
    with np.load(os.path.join(HERE, "cfg2_xyz13_28_ani2x.npz")) as z:
        old = [str(s) for s in z["symbols"]]
        sp, x = z["species"][:12], z["coords"][:12]
    remap = np.asarray([SYM.index(s) if s in SYM else -9 for s in old] + [-1])
    species = remap[sp]
    assert species.min() >= -1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = simple_ani(SYM, "wb97x-631gd", periodic_table_index=False, **c["kw"])
    in_dim = model.aev_computer.out_dim
    hidden = {s: c["hidden"][s] for s in SYM}
    sd = {k: torch.from_numpy(v) for k, v in
          random_network_state_dict(SYM, in_dim, hidden, c["kw"]["ensemble_size"], c["seed"], c["bias"]).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if "neural_networks" in k], (missing[:3], unexpected[:3])
    model = model.double()
    elem = torch.from_numpy(species.astype(np.int64))
    coords = torch.from_numpy(x).double().requires_grad_(True)
    e = model((elem, coords)).energies
    (grad,) = torch.autograd.grad(e.sum(), coords)
    path = os.path.join(HERE, f"simple_{case}.npz")
    np.savez_compressed(path, symbols=np.asarray(SYM), species=species.astype(np.int64), coords=x, seed=np.asarray(c["seed"]),
                        energies=e.detach().numpy(), forces=(-grad).numpy(), aev_dim=np.asarray(in_dim),
                        self_energies=model.energy_shifter.self_energies.detach().numpy())
    print(f"{path}: aev {in_dim}  E[0]={e[0].item():+.9f} |F|max={grad.abs().max().item():.5f}")


def run_q(seed=43):
    """simple_aniq (arch.py:1069-1185): separate charge networks + normalizer; charges and dipoles (electro.py)."""
    from torchani.arch import simple_aniq
    from torchani.electro import compute_dipole

    with np.load(os.path.join(HERE, "cfg2_xyz13_28_ani2x.npz")) as z:
        old = [str(s) for s in z["symbols"]]
        sp, x = z["species"][:12], z["coords"][:12]
    remap = np.asarray([SYM.index(s) if s in SYM else -9 for s in old] + [-1])
    species = remap[sp]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = simple_aniq(SYM, "wb97x-631gd", ensemble_size=2, periodic_table_index=False)
    in_dim = model.aev_computer.out_dim
    hidden = {s: HIDDEN_DIMS_2X[s] for s in SYM}
    sd = {k: torch.from_numpy(v) for k, v in random_network_state_dict(SYM, in_dim, hidden, 2, seed, False).items()}
    pre = "potentials.nnp.neural_networks."
    sd.update({"potentials.nnp.charge_networks." + k[len(pre):]: torch.from_numpy(v)
               for k, v in random_network_state_dict(SYM, in_dim, hidden, 1, 1000 + seed, False, scale=3.0).items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if "networks" in k], (missing[:3], unexpected[:3])
    model = model.double()
    elem = torch.from_numpy(species.astype(np.int64))
    coords = torch.from_numpy(x).double()
    out = model((elem, coords))
    znum = torch.tensor([1, 6, 7, 8, -1])[elem]      # (-1 stays -1)
    dip = {ref: compute_dipole(znum, coords, out.atomic_charges.detach(), ref).numpy()
           for ref in ("center_of_mass", "center_of_geometry", "origin")}
    path = os.path.join(HERE, "simple_chnoq.npz")
    np.savez_compressed(path, symbols=np.asarray(SYM), species=species.astype(np.int64), coords=x, seed=np.asarray(seed),
                        energies=out.energies.detach().numpy(), atomic_charges=out.atomic_charges.detach().numpy(),
                        atomic_numbers=znum.numpy(), **{"dipole_" + k: v for k, v in dip.items()})
    q = out.atomic_charges.detach().numpy()
    print(f"{path}: E[0]={out.energies[0].item():+.9f} |q|max={np.abs(q).max():.4f} |mu|max={np.abs(dip['center_of_mass']).max():.4f}")


if __name__ == "__main__":
    for case in CASES:
        run(case)
    run_q()
