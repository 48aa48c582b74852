"""GPU tests of the Verlet-skin neighbor reuse (anihip_nbr_refresh; the reference's VerletCellList,
neighbors.py:759-884) and of the MD driver built on it (the reference drives the same path through
ASE, torchani/ase.py + tools/md-benchmark.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from _util import STRESS_NAMES, load_golden, load_stress, oracle_networks, oracle_params, seeded_state
from test_gpu_parity import report

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from torchani_amd import _lib

    _lib.lib()  # fail loudly if the native library is missing
    return torch.device("cuda:0")


def water(side):
    from bench import water_box

    sp, x, cell = water_box(side)
    return sp.astype(np.int32), x.astype(np.float32), cell.astype(np.float32), (True, True, True)


def cases():
    out = {}
    sp, x, cell, pbc = water(10)
    out["water3000_cell"] = (sp, x, cell, pbc, "cell")
    g = load_golden("triclinic_pbc_ani2x")
    out["triclinic_batch"] = (g["species"].astype(np.int32), g["coords"], g["cell"], tuple(bool(b) for b in g["pbc"]),
                              "batch")
    g = load_golden("rand_batch_ani2x")
    out["rand_batch"] = (g["species"].astype(np.int32), g["coords"], None, None, "batch")
    return out


@pytest.mark.parametrize("name", ["water3000_cell", "triclinic_batch", "rand_batch"])
def test_verlet_refresh_equals_rebuild(dev, name):
    from torchani_amd.constants import aev_constants_2x
    from torchani_amd.engine import AevEngine, VerletRows

    sp, x, cell, pbc, mode = cases()[name]
    eng = AevEngine(aev_constants_2x())
    spd = torch.from_numpy(sp).to(dev)
    x0 = torch.from_numpy(x).to(dev).contiguous()
    cd = None if cell is None else torch.from_numpy(cell).to(dev)
    n = spd.numel()
    ver = VerletRows(skin=1.0)
    ver.rebuild_above = float("inf")   # (the reuse machinery: by default the list is an API shim that rebuilds every step)
    r0 = ver.rows(eng, spd, x0, cd, pbc, 0, n, mode, 256)
    f0 = eng.neighbors(spd, x0, cd, pbc, mode=mode, row_cap=256)
    assert torch.equal(r0.meta[:, 1:], f0.meta[:, 1:])      # same counts per class and species at the build point
    rs = np.random.RandomState(3)
    # every atom moves by less than skin / 2 = 0.5 A (|delta| <= 0.28 sqrt(3) = 0.485)
    x1 = (x0 + torch.from_numpy(rs.uniform(-0.28, 0.28, x.shape).astype(np.float32)).to(dev)).contiguous()
    r1 = ver.rows(eng, spd, x1, cd, pbc, 0, n, mode, 256)
    assert (ver.n_builds, ver.n_reuses) == (1, 1)
    f1 = eng.neighbors(spd, x1, cd, pbc, mode=mode, row_cap=256)
    assert int(r1.status[0]) & 3 == 0 and int(f1.status[0]) & 3 == 0
    # identical neighbor sets: counts per (class, species) match exactly; the AEVs (sums over the rows) to round-off
    same = torch.equal(r1.meta[:, 1:], f1.meta[:, 1:])
    a1, b1 = eng.forward(spd, r1), eng.forward(spd, f1)
    err = (a1 - b1).abs().max().item()
    report(f"verlet {name:16s} refresh vs rebuild: counts equal = {same}  max|aev diff| = {err:.2e}")
    assert same and err < 2e-5
    # the backward pass through refreshed rows
    w = torch.from_numpy(rs.uniform(-1, 1, tuple(a1.shape)).astype(np.float32)).to(dev)
    ga, gb = eng.backward(spd, r1, w), eng.backward(spd, f1, w)
    assert (ga - gb).abs().max().item() < 1e-4 * max(1.0, gb.abs().max().item())
    # one atom jumps beyond skin / 2: the pair search must run again
    x2 = x1.clone()
    x2.view(-1, 3)[0, 0] += 0.9
    ver.rows(eng, spd, x2, cd, pbc, 0, n, mode, 256)
    assert ver.n_builds == 2
    # sharded refresh: rows of a sub-range equal the rows of the full call
    lo, hi = n // 3, n // 3 + max(1, n // 4)
    rpart = eng.refresh_rows(spd, x2, ver._coords0, ver._rows, lo, hi, 256)
    rfull = eng.refresh_rows(spd, x2, ver._coords0, ver._rows, 0, n, 256)
    assert torch.equal(rpart.meta[lo:hi, 1:], rfull.meta[lo:hi, 1:])


def make_model(dev, neighborlist):
    from torchani_amd.models import ANI2x

    return ANI2x(state_dict=seeded_state("ani2x", 8, 7), device=dev, periodic_table_index=False,
                 neighborlist=neighborlist)


def test_md_verlet_matches_plain_and_conserves_energy(dev):
    from torchani_amd.md import MolecularDynamics

    sp, x, cell, pbc = water(10)
    spd = torch.from_numpy(sp.astype(np.int64)).to(dev)
    xd = torch.from_numpy(x).to(dev)
    cd = torch.from_numpy(cell).to(dev)
    masses = torch.tensor([1.008, 12.011, 14.007, 15.999, 32.06, 18.998, 35.45], device=dev)[spd]
    runs = {}
    for nl in ("cell_list", "verlet_cell_list"):
        model = make_model(dev, nl)
        if model.aev_computer.verlet is not None:
            model.aev_computer.verlet.rebuild_above = float("inf")   # (exercise the reuse path, off by default)
        md = MolecularDynamics(model, spd, xd, cd, pbc, dt=0.25, masses=masses, seed=1)
        md.set_temperature(150.0)
        e0 = md.total_energies().clone()
        ke0 = md.kinetic_energies().clone()
        md.run(60)
        torch.cuda.synchronize()
        runs[nl] = (md, e0, ke0)
    md_p, e0, ke0 = runs["cell_list"]
    md_v = runs["verlet_cell_list"][0]
    ver = md_v.model.aev_computer.verlet
    drift = (md_v.total_energies() - e0).abs().item()
    dx = (md_v.coords - md_p.coords).abs().max().item()
    moved = (md_v.coords - xd).norm(dim=-1).max().item()
    dke = (md_v.kinetic_energies() - ke0).abs().item()
    report(f"md     water 3000 atoms, 60 x 0.25 fs NVE: |dE_total| = {drift:.2e} Ha (|dKE| = {dke:.2e}, KE0 = "
           f"{ke0.item():.3f}), max move {moved:.3f} A, verlet builds/reuses = {ver.n_builds}/{ver.n_reuses}, "
           f"max|x_verlet - x_plain| = {dx:.2e} A")
    assert ver.n_reuses > 0 and ver.n_builds >= 1
    assert dx < 1e-3
    assert drift < 0.02 * max(dke, 1e-3) + 1e-4


@pytest.mark.parametrize("base", STRESS_NAMES)
def test_virial_matches_reference_stress(dev, base):
    """energies_and_forces(stress=True): the fdotr virial of the HIP backward against the reference's scaling stress
    (tests/golden/gen_golden_stress.py).  Tolerance like the forces: 1e-4 Ha per unit strain (measured ~1e-7)."""
    from torchani_amd.models import ANI2x

    g, st = load_golden(base), load_stress(base)
    for mode in ("batch", "cell"):
        model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False,
                      neighborlist=mode, row_capacity=256, cutoff_fn=g["cutoff_fn"])
        sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
        x = torch.from_numpy(g["coords"]).to(dev)
        cell = torch.from_numpy(g["cell"]).to(dev)
        pbc = tuple(bool(b) for b in g["pbc"])
        out = model.energies_and_forces(sp, x, cell, pbc, stress=True)
        torch.cuda.synchronize()
        vir = out.virial.cpu().numpy()
        err = np.abs(vir - st["virial"]).max()
        report(f"stress {base:24s} {mode:5s} max|virial err| = {err:.2e} Ha (max |virial| {np.abs(st['virial']).max():.3f})")
        assert err < 1e-4
        assert np.abs(vir - vir.T).max() == 0.0
        # forces are those of the plain call, and two half shards add up to the whole virial
        plain = model.energies_and_forces(sp, x, cell, pbc)
        assert plain.virial is None and torch.equal(plain.energies, out.energies)
        parts = [model.energies_and_forces(sp, x, cell, pbc, shard=(r, 2), stress=True).virial for r in range(2)]
        assert (parts[0] + parts[1] - out.virial).abs().max().item() < 1e-6


def test_virial_finite_difference_large_box(dev):
    """Strain derivative by central differences of the energy on the 3000-atom water box (size-independent property:
    no oracle needed): dE/d eps_ab = virial_ab."""
    sp, x, cell, pbc = water(10)
    model = make_model(dev, "cell_list")
    spd = torch.from_numpy(sp.astype(np.int64)).to(dev)
    xd = torch.from_numpy(x).to(dev).double()
    cd = torch.from_numpy(cell).to(dev).double()
    out = model.energies_and_forces(spd, xd.float(), cd.float(), pbc, stress=True)
    h = 2e-3
    for (a, b) in ((0, 0), (1, 2)):
        es = []
        for sgn in (+1.0, -1.0):
            eps = torch.eye(3, dtype=torch.float64, device=dev)
            eps[a, b] += sgn * h / 2
            eps[b, a] += sgn * h / 2
            o = model.energies_and_forces(spd, (xd @ eps).float(), (cd @ eps).float(), pbc)
            es.append(o.energies.item())
        fd = (es[0] - es[1]) / (2 * h)
        v = 0.5 * (out.virial[a, b] + out.virial[b, a]).item()
        report(f"stress water3000 d E/d eps[{a}{b}]: finite difference {fd:+.5f}  virial {v:+.5f}")
        assert abs(fd - v) < 2e-3 * max(1.0, abs(v))


class FakeAtoms:
    """The slice of ase.Atoms the calculator touches."""

    def __init__(self, numbers, positions, cell, pbc):
        self.numbers, self.positions, self.cell, self.pbc = numbers, positions.copy(), cell, pbc

    def get_atomic_numbers(self):
        return self.numbers

    def get_positions(self):
        return self.positions

    def set_positions(self, p):
        self.positions = np.asarray(p, dtype=np.float64)

    def get_cell(self, complete=True):
        return self.cell

    def get_pbc(self):
        return self.pbc

    def get_volume(self):
        return abs(np.linalg.det(self.cell))


def test_ase_calculator_protocol(dev):
    """model.ase().calculate(atoms, ["energy", "forces", "stress"]) in ASE units (ase.py:75-173)."""
    from torchani_amd.ase import HARTREE_TO_EV
    from torchani_amd.models import ANI2x

    g, st = load_golden("water_pbc_ani2x"), load_stress("water_pbc_ani2x")
    model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev)   # periodic_table_index=True
    znum = np.asarray([1, 6, 7, 8, 16, 9, 17])[g["species"][0]]
    shifted = g["coords"][0].astype(np.float64) + 3.0 * g["cell"][0]           # outside the central cell
    atoms = FakeAtoms(znum, shifted, g["cell"].astype(np.float64), np.asarray(g["pbc"]))
    calc = model.ase(overwrite=True)
    calc.calculate(atoms, ["energy", "forces", "stress"])
    assert abs(calc.results["energy"] - g["energies"][0] * HARTREE_TO_EV) < 1e-4
    assert calc.results["free_energy"] == calc.results["energy"]
    assert np.abs(calc.results["forces"] - g["forces"][0] * HARTREE_TO_EV).max() < 1e-4 * HARTREE_TO_EV
    vol = abs(np.linalg.det(g["cell"].astype(np.float64)))
    assert np.abs(calc.results["stress"] - st["virial"] * HARTREE_TO_EV / vol).max() < 1e-6
    # overwrite=True wrote the wrapped positions back (ase.py:101-104)
    frac = atoms.get_positions() @ np.linalg.inv(g["cell"].astype(np.float64))
    assert frac.min() > -1e-5 and frac.max() < 1 + 1e-5
    with pytest.raises(ValueError, match="periodic_table_index"):
        ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False).ase()


def test_verlet_list_rebuilds_by_default(dev):
    """neighborlist="verlet_cell_list" is an API-compatibility shim (engine.VerletRows): by default every step takes the plain
    pair search -- measured faster than or equal to the skin refresh at every size -- and gives the cell list's results."""
    sp, x, cell, pbc = water(10)
    spd = torch.from_numpy(sp.astype(np.int64)).to(dev)
    xd, cd = torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
    out = {}
    for nl in ("cell_list", "verlet_cell_list"):
        model = make_model(dev, nl)
        out[nl] = model.energies_and_forces(spd, xd, cd, pbc)
        if nl == "verlet_cell_list":
            ver = model.aev_computer.verlet
            assert ver.rebuild_above == 0 and ver.n_direct >= 1 and ver.n_builds == 0 and ver.n_reuses == 0
    assert torch.equal(out["cell_list"].energies, out["verlet_cell_list"].energies)
    assert float((out["cell_list"].forces - out["verlet_cell_list"].forces).abs().max()) < 2e-6
