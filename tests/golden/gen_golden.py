"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself (fp64, pyaev).

Run only where /root/reference exists (the build container); the resulting *.npz are committed so that
the GPU box -- where the reference is absent -- can check the oracle and the HIP engine against them.

    python tests/golden/gen_golden.py

The reference model is the ANI-2x / ANI-1x *architecture* (Assembler, arch.py:742-989, recipe of
models.py:185-193 without the download) loaded with torchani_amd.weights.random_state_dict(kind, 8, seed),
i.e. the same seeded parameters the tests re-create on the GPU box.  Everything is evaluated in float64
(``model.double()``): fp32-rounded parameters and AEV constants in fp64 arithmetic.
Inputs are rounded to fp32 first so that the fp32 engine sees bit-identical coordinates.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

os.environ["TORCHANI_NO_WARN_EXTENSIONS"] = "1"


class _Any:
    def __class_getitem__(cls, k):
        return cls


for _name in ("h5py", "zarr"):
    _m = types.ModuleType(_name)
    for _k in ("File", "Group", "Dataset", "Datatype"):
        setattr(_m, _k, type(_k, (_Any,), {}))
    sys.modules[_name] = _m
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import torchani  # noqa: E402
from torchani.arch import Assembler  # noqa: E402
from torchani.io import read_xyz  # noqa: E402
from torchani.utils import SYMBOLS_1X, SYMBOLS_2X  # noqa: E402

from torchani_amd.weights import random_state_dict  # noqa: E402

RES = "/root/reference/tests/resources"
CUTOFF_FN = "cosine"   # gen_golden_smooth.py switches this to "smooth" (cutoffs.py:84-101)
ZNUM = {"H": 1, "C": 6, "N": 7, "O": 8, "S": 16, "F": 9, "Cl": 17}


def build_reference(kind: str, seed: int, neighborlist: str = "all_pairs"):
    asm = Assembler()
    asm.set_symbols(SYMBOLS_2X if kind == "ani2x" else SYMBOLS_1X)
    asm.set_global_cutoff_fn(CUTOFF_FN)
    asm.set_aev_computer(radial=kind, angular=kind, strategy="pyaev")
    asm.set_atomic_networks(ctor=kind)
    asm.set_neighborlist(neighborlist)
    asm.set_gsaes_as_self_energies("wb97x-631gd")
    model = asm.assemble(8)
    sd = model.state_dict()
    mine = random_state_dict(kind, 8, seed)
    for k, v in mine.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model.requires_grad_(False)
    return model.double()


def run_case(name, kind, seed, znums, coords, cell=None, pbc=None, aev_rows=None,
             neighborlist="all_pairs"):
    model = build_reference(kind, seed, neighborlist)
    symbols = SYMBOLS_2X if kind == "ani2x" else SYMBOLS_1X
    znums = torch.as_tensor(znums, dtype=torch.long)
    coords32 = torch.as_tensor(np.asarray(coords), dtype=torch.float32)
    coords = coords32.double().requires_grad_(True)
    cell_t = None if cell is None else torch.as_tensor(np.asarray(cell), dtype=torch.float32).double()
    pbc_t = None if pbc is None else torch.as_tensor(pbc, dtype=torch.bool)
    elem = model.species_converter(znums)
    aev = model.aev_computer(elem, coords, cell_t, pbc_t)
    # NN-only per-atom energies (energy shifter off), then totals with the shifter on
    model.set_enabled("energy_shifter", False)
    atomic = model((znums, coords), cell_t, pbc_t, atomic=True).energies
    e_nn = atomic.sum(dim=1)
    (grad,) = torch.autograd.grad(e_nn.sum(), coords)
    model.set_enabled("energy_shifter", True)
    e_tot = model((znums, coords), cell_t, pbc_t).energies
    members = None
    if name in ("simple2_ani2x",):
        model.set_enabled("energy_shifter", False)
        members = model((znums, coords), cell_t, pbc_t, atomic=True, ensemble_values=True).energies
        model.set_enabled("energy_shifter", True)
    # vector-Jacobian product of the AEV alone with seeded cotangents: pins the AEV backward without
    # going through the networks (torch's fp64 celu backward carries a ~1e-8 relative artefact: it
    # evaluates exp(x * (1/float32(alpha))), see tests/test_oracle_golden.py)
    w = np.random.RandomState(seed + 1000).uniform(-1.0, 1.0, tuple(aev.shape))
    (vjp,) = torch.autograd.grad((aev * torch.as_tensor(w)).sum(), coords, retain_graph=True)
    aev_np = aev.detach().numpy()
    n_flat = aev_np.shape[0] * aev_np.shape[1]
    rows = np.arange(n_flat) if aev_rows is None else np.asarray(aev_rows)
    out = dict(
        kind=np.asarray(kind), seed=np.asarray(seed), n_members=np.asarray(8),
        symbols=np.asarray(symbols), species=elem.numpy().astype(np.int32),
        coords=coords32.numpy(), aev_rows=rows.astype(np.int64),
        aev=aev_np.reshape(n_flat, -1)[rows], atomic_energies=atomic.detach().numpy(),
        energies_nn=e_nn.detach().numpy(), energies=e_tot.detach().numpy(),
        forces=(-grad).numpy(), aev_vjp=vjp.numpy(),
    )
    if cell is not None:
        out["cell"] = np.asarray(cell, dtype=np.float32)
        out["pbc"] = np.asarray(pbc, dtype=bool)
    if members is not None:
        out["member_atomic_energies"] = members.detach().numpy()
    if CUTOFF_FN != "cosine":
        out["cutoff_fn"] = np.asarray(CUTOFF_FN)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: C={elem.shape[0]} A={elem.shape[1]} E_nn[0]={e_nn[0].item():+.9f} "
          f"E[0]={e_tot[0].item():+.6f} |F|max={grad.abs().max().item():.4f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def random_molecules(rs, n_mol, n_atoms, n_species, box, min_dist, pad_prob=0.0):
    """Rejection-sampled coordinates in a cube; trailing atoms padded with probability pad_prob."""
    zs = [1, 6, 7, 8, 16, 9, 17][:n_species]
    Z = np.full((n_mol, n_atoms), -1, dtype=np.int64)
    X = np.zeros((n_mol, n_atoms, 3), dtype=np.float32)
    for m in range(n_mol):
        n_real = n_atoms if m == 0 else max(1, n_atoms - rs.binomial(n_atoms, pad_prob))
        pts = []
        while len(pts) < n_real:
            p = rs.uniform(0, box, 3)
            if all(np.linalg.norm(p - q) >= min_dist for q in pts):
                pts.append(p)
        X[m, :n_real] = np.asarray(pts, dtype=np.float32)
        Z[m, :n_real] = rs.choice(zs, n_real)
    return Z, X


def xyz(path):
    z, x, cell, pbc = read_xyz(path)
    return z.numpy(), x.numpy(), (None if cell is None else cell.numpy()), (
        None if pbc is None else pbc.numpy())


def main():
    torch.set_num_threads(8)
    # 1. config 1: methane, ANI-1x architecture (tests/resources/CH4-5.xyz first conformer)
    z, x, _, _ = xyz(f"{RES}/CH4-5.xyz")
    run_case("ch4_ani1x", "ani1x", 11, z[:1], x[:1])

    # 2. two molecules, one of them a lone atom + padding (isolated-atom and padding edge cases,
    #    cf. tests/test_aev.py:61-131, tests/test_cuaev.py:301-326)
    Z = np.array([[6, 1, 1, 1, 1, 8, 1], [7, -1, -1, -1, -1, -1, -1]])
    X = np.zeros((2, 7, 3), dtype=np.float32)
    X[0] = [[0.03, 0.006, 0.01], [-0.83, 0.39, -0.26], [-0.67, -0.85, 0.31], [0.46, 0.54, 0.82],
            [0.66, -0.17, -0.91], [2.9, 0.4, 0.2], [3.3, 1.2, -0.1]]
    X[1, 0] = [10.0, 10.0, 10.0]
    run_case("simple2_ani2x", "ani2x", 12, Z, X)

    # 3. random padded batch over all 7 species (cf. tests/test_cuaev.py:337-351)
    rs = np.random.RandomState(3)
    Z, X = random_molecules(rs, 6, 14, 7, 4.2, 0.75, pad_prob=0.3)
    run_case("rand_batch_ani2x", "ani2x", 13, Z, X)

    # 4. periodic water box, 8 A cube (tests/resources/water-0.8nm.xyz; tests/test_cuaev.py:675-718)
    z, x, cell, pbc = xyz(f"{RES}/water-0.8nm.xyz")
    run_case("water_pbc_ani2x", "ani2x", 14, z, x, cell, pbc)

    # 5. triclinic cell thinner than the cutoff along one axis, mixed pbc (many images per pair)
    rs = np.random.RandomState(5)
    cell = np.array([[6.3, 0.0, 0.0], [1.7, 4.1, 0.0], [0.9, -1.1, 9.0]], dtype=np.float32)
    frac = rs.uniform(-0.4, 1.4, (1, 20, 3))  # some atoms start outside the cell -> wrapping
    frac[..., 2] = rs.uniform(0.2, 0.8, (1, 20))
    X = (frac @ cell.astype(np.float64)).astype(np.float32)
    Z = rs.choice([1, 6, 7, 8, 16, 9, 17], (1, 20))
    run_case("triclinic_pbc_ani2x", "ani2x", 15, Z, X, cell, [True, True, False])

    # 6. small protein fragment, 264 atoms, no pbc (tests/resources/small.xyz; tests/test_cuaev.py:637-673)
    z, x, _, _ = xyz(f"{RES}/small.xyz")
    run_case("small_ani2x", "ani2x", 16, z, x, aev_rows=np.arange(0, 264, 6))

    # 7. 1hz5 protein, 973 atoms (drives the cell-list path without pbc)
    z, x, _, _ = xyz(f"{RES}/1hz5.xyz")
    run_case("1hz5_ani2x", "ani2x", 17, z, x, aev_rows=np.arange(0, 973, 31))

    # 8. very dense cluster: >64 angular neighbors per atom (cf. dense case tests/test_cuaev.py:593-635)
    rs = np.random.RandomState(8)
    Z, X = random_molecules(rs, 1, 90, 7, 3.6, 0.55)
    run_case("dense90_ani2x", "ani2x", 18, Z, X, aev_rows=np.arange(0, 90, 9))

    # 9. benzene in a periodic cell via the cell list of the reference (tests/test_aev.py:390-436 input)
    z, x, cell, pbc = xyz(f"{RES}/benzene.xyz")
    if cell is not None:
        run_case("benzene_pbc_ani2x", "ani2x", 19, z, x, cell, pbc, neighborlist="cell_list")


if __name__ == "__main__":
    main()
