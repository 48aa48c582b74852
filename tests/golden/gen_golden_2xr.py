"""Golden fixtures of the ANI-2xr / ANI-2dr ARCHITECTURE from the reference's own builder (arch.py:992-1066 simple_ani,
recipes of models.py:252-325): AEV with the smooth envelope, GELU networks without biases, xTB repulsion, and for -2dr
the DFT-D3(BJ) dispersion term and the B97-3c self energies -- with torchani_amd.weights.random_state_dict(kind, 8, seed)
as parameters (the published ones are a download), everything in float64:

    python tests/golden/gen_golden_2xr.py     -> tests/golden/x2r_<kind>_<case>.npz

Coordinates / elements of existing golden cases; element indices are re-expressed in the models' atomic-number order
(H C N O F S Cl).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_d3 as g3  # noqa: E402  (reference import + h5py stand-in)
import gen_golden as gg  # noqa: E402

import torch  # noqa: E402
from torchani.arch import simple_ani  # noqa: E402

from torchani_amd.weights import arch_spec, random_state_dict  # noqa: E402

LOT = {"ani2xr": "wb97x-631gd", "ani2dr": "b973c-def2mtzvp", "anir2s": "r2scan3c-def2mtzvpp"}
# models.py:325-368: the ANI-2x AEV with the smooth envelope, repulsion without a cutoff
R2S_KW = dict(repulsion_cutoff=False, cutoff_fn="smooth", radial_start=0.8, angular_start=0.8, radial_cutoff=5.1)


def run(kind, name, seed):
    with np.load(os.path.join(HERE, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = arch_spec(kind)[0]
    old = [str(s) for s in g["symbols"]]
    remap = np.asarray([symbols.index(s) for s in old] + [-1])
    species = remap[g["species"]]          # (-1 stays -1)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = simple_ani(lot=LOT[kind], symbols=symbols, ensemble_size=8, dispersion=kind == "ani2dr", repulsion=True,
                           periodic_table_index=False, **(R2S_KW if kind == "anir2s" else {}))
    state = {k: torch.from_numpy(v) for k, v in random_state_dict(kind, 8, seed).items()}
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not [k for k in missing if "neural_networks" in k or "energy_shifter" in k], missing[:3]
    assert not unexpected, unexpected[:3]
    model = model.double()
    elem = torch.from_numpy(species.astype(np.int64))
    coords = torch.from_numpy(g["coords"]).double().requires_grad_(True)
    cell = torch.from_numpy(g["cell"]).double() if "cell" in g else None
    pbc = torch.from_numpy(g["pbc"]) if "pbc" in g else None
    e = model((elem, coords), cell, pbc).energies
    (grad,) = torch.autograd.grad(e.sum(), coords)
    out = dict(kind=np.asarray(kind), seed=np.asarray(seed), symbols=np.asarray(symbols), species=species.astype(np.int64),
               coords=g["coords"], energies=e.detach().numpy(), forces=(-grad).numpy())
    if "cell" in g:
        out["cell"], out["pbc"] = g["cell"], g["pbc"]
    path = os.path.join(HERE, f"x2r_{kind}_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: E[0]={e[0].item():+.9f} |F|max={grad.abs().max().item():.5f}")


if __name__ == "__main__":
    only = sys.argv[1:]
    for kind in ("ani2xr", "ani2dr"):
        for nm, seed in (("rand_batch_ani2x", 21), ("water_pbc_ani2x", 22), ("small_ani2x", 23)):
            if not only or kind in only:
                run(kind, nm, seed)
    for nm, seed in (("rand_batch_ani2x", 24), ("dense90_ani2x", 25)):   # (molecules: the repulsion has no cutoff)
        if not only or "anir2s" in only:
            run("anir2s", nm, seed)
