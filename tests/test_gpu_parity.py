"""GPU parity tests: the HIP engine (through the C ABI) against the golden fixtures written by the
reference and against the pinned CPU oracle, on identical inputs.

Tolerances (north_star): per-atom / NN-only energies <= 1e-5 Ha, forces <= 1e-4 Ha/A; AEV elements are
held to 2e-5 absolute (the reference's own cuAEV-vs-pyaev gate is 5e-5, tests/test_cuaev.py:169).
Modelled on tests/test_cuaev.py:152-798 of the reference (differential tests of the native path).
"""
import os
import sys

import numpy as np
import pytest
import torch

from _util import GOLDEN_NAMES, load_golden, oracle_networks, oracle_params, seeded_state
from torchani_amd import _lib
from torchani_amd.engine import PackedNetworks

pytestmark = pytest.mark.gpu

# north_star's gates (BASELINE.json): what "matches the reference" means
AEV_TOL = 2e-5
E_ATOM_TOL = 1e-5
F_TOL = 1e-4
# REGRESSION gates, ~20x what the engine measures on these cases (per-atom energies <= 7e-8 Ha, forces <= 2.5e-7 Ha/A,
# AEV vector-Jacobian products <= 2e-6 of the largest entry): a kernel bug that stays inside the parity gates -- round 3's
# missing wait state gave 2 % wrong dE/dAEV, |dF| = 9e-5 -- must fail HERE.  (The reference gates its own cuAEV at its
# noise floor too: tests/test_cuaev.py:169.)
E_ATOM_REG = 1e-6
F_REG = 5e-6
VJP_REG_REL = 5e-6
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                      "parity_report.txt")


def report(line):
    print(line)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from torchani_amd import _lib

    _lib.lib()  # fail loudly if the native library is missing
    return torch.device("cuda:0")


_models = {}


def get_model(kind, seed, dev, **kw):
    from torchani_amd.models import ANI1x, ANI2x

    key = (kind, seed, tuple(sorted(kw.items())))
    if key not in _models:
        ctor = ANI2x if kind == "ani2x" else ANI1x
        _models[key] = ctor(state_dict=seeded_state(kind, 8, seed), device=dev, periodic_table_index=False, **kw)
    return _models[key]


def modes_for(g):
    return ["batch", "cell"] if g["species"].shape[0] == 1 else ["batch"]


def to_dev(g, dev):
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    cell = None if g["cell"] is None else torch.from_numpy(g["cell"]).to(dev)
    pbc = None if g["pbc"] is None else tuple(bool(b) for b in g["pbc"])
    return sp, x, cell, pbc


def unpack_rows(nbrs, n):
    meta = nbrs.meta.cpu().numpy().view(np.uint32).reshape(n, 6)
    ent = nbrs.ent.cpu().numpy().reshape(-1, 4)
    return meta, ent


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_neighbor_rows_match_oracle(dev, oracle64, name):
    """Same neighbor set per atom as the reference's pair list (seen from both ends), correct
    angular/far split, species ordering and per-species counts."""
    from torchani_amd.engine import AevEngine
    from torchani_amd.weights import arch_spec

    g = load_golden(name)
    consts = arch_spec(g["kind"])[1]._replace(cutoff_fn=g["cutoff_fn"])
    eng = AevEngine(consts)
    sp, x, cell, pbc = to_dev(g, dev)
    C, A = g["species"].shape
    start, j, d, r = oracle64.neighbors(g["species"], g["coords"].astype(np.float64), consts.Rcr, g["cell"],
                                        g["pbc"])
    for mode in modes_for(g):
        nbrs = eng.neighbors(sp.to(torch.int32).contiguous(), x.contiguous(), cell, pbc, mode=mode, row_cap=256)
        torch.cuda.synchronize()
        nbrs.raise_on_overflow()
        meta, ent = unpack_rows(nbrs, C * A)
        worst = 0.0
        for i in range(C * A):
            nA, nF = int(meta[i, 1] & 0xFFFF), int(meta[i, 1] >> 16)
            lo, hi = start[i], start[i + 1]
            assert nA + nF == hi - lo, f"{name}/{mode}: atom {i} has {nA + nF} neighbors, oracle {hi - lo}"
            if hi == lo:
                continue
            row = ent[meta[i, 0]: meta[i, 0] + nA + nF]
            w = row[:, 3].copy().view(np.uint32)
            jj, spj = (w & 0x0FFFFFFF).astype(np.int64), (w >> 28).astype(np.int64)
            assert np.array_equal(spj, g["species"].reshape(-1)[jj])
            rr = np.linalg.norm(row[:, :3].astype(np.float64), axis=1)
            # angular-range group first, each group sorted by species
            assert np.all(rr[:nA] <= consts.Rca + 1e-5) and np.all(rr[nA:] >= consts.Rca - 1e-5)
            assert np.all(np.diff(spj[:nA]) >= 0) and np.all(np.diff(spj[nA:]) >= 0)
            cntA = np.concatenate([(meta[i, 2] >> (8 * np.arange(4))) & 255, (meta[i, 3] >> (8 * np.arange(4))) & 255])
            cntF = np.concatenate([(meta[i, 4] >> (8 * np.arange(4))) & 255, (meta[i, 5] >> (8 * np.arange(4))) & 255])
            assert np.array_equal(cntA, np.bincount(spj[:nA], minlength=8))
            assert np.array_equal(cntF, np.bincount(spj[nA:], minlength=8))
            # same (j, displacement) multiset as the oracle
            ko = np.lexsort((d[lo:hi, 2], d[lo:hi, 1], d[lo:hi, 0], j[lo:hi]))
            km = np.lexsort((row[:, 2], row[:, 1], row[:, 0], jj))
            assert np.array_equal(j[lo:hi][ko], jj[km]), f"{name}/{mode}: atom {i} neighbor indices differ"
            worst = max(worst, np.abs(d[lo:hi][ko] - row[km, :3]).max())
        report(f"nbr   {name:22s} {mode:5s} max|d - d_ref| = {worst:.2e} A")
        assert worst < 5e-6


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_aev_forward_and_backward(dev, name):
    from torchani_amd.aev import AEVComputer
    from torchani_amd.weights import arch_spec

    g = load_golden(name)
    consts = arch_spec(g["kind"])[1]._replace(cutoff_fn=g["cutoff_fn"])
    sp, x, cell, pbc = to_dev(g, dev)
    C, A = g["species"].shape
    pbc_t = None if pbc is None else torch.tensor(pbc)
    w = torch.from_numpy(np.random.RandomState(g["seed"] + 1000).uniform(-1.0, 1.0, (C, A, consts.out_dim))
                         .astype(np.float32)).to(dev)
    for mode in modes_for(g):
        aevc = AEVComputer(consts, neighborlist=mode, row_capacity=256).to(dev)
        xx = x.clone().requires_grad_(True)
        aev = aevc(sp, xx, cell, pbc_t)
        (vjp,) = torch.autograd.grad((aev * w).sum(), xx)
        torch.cuda.synchronize()
        aevc.last_neighbors().raise_on_overflow()
        got = aev.detach().cpu().numpy().reshape(C * A, -1)
        err = np.abs(got[g["aev_rows"]] - g["aev"]).max()
        pad = g["species"].reshape(-1) < 0
        assert np.all(got[pad] == 0), "padding atoms must have zero AEV rows"
        verr = np.abs(vjp.cpu().numpy() - g["aev_vjp"]).max()
        vmag = np.abs(g["aev_vjp"]).max()
        report(f"aev   {name:22s} {mode:5s} max|aev err| = {err:.2e}   vjp err = {verr:.2e} (|vjp|max {vmag:.1f})")
        assert err < AEV_TOL
        assert verr < 2e-5 * max(1.0, vmag)
        assert verr <= VJP_REG_REL * max(1.0, vmag), "regression gate (module header): AEV backward"
        assert np.all(vjp.cpu().numpy()[g["species"] < 0] == 0)


GRID_CASES = ["r8_a4z4_batch", "r24_a10z8_pbc", "r5_a3z5_dense"]


@pytest.mark.parametrize("name", GRID_CASES)
def test_general_symmetry_function_grids(dev, name):
    """AEVComputer.from_constants with grids other than 16 / 8x4 / 4x8 (csrc/aev_generic.hip; the reference's templated
    cuAEV kernels, csrc/aev.cu:1687-1777): AEV rows and the coordinate gradient of a seeded linear functional against the
    reference's own numbers (tests/golden/gen_golden_grids.py); the virial and the fixed-point accumulation against the
    float path; a sharded backward (central atoms lo..hi) adds up to the whole."""
    from torchani_amd.aev import AEVComputer

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"grid_{name}.npz")) as z:
        g = {k: z[k] for k in z.files}
    aevc = AEVComputer.from_constants(float(g["Rcr"]), float(g["Rca"]), float(g["EtaR"]), g["ShfR"].tolist(), float(g["EtaA"]),
                                      float(g["Zeta"]), g["ShfA"].tolist(), g["ShfZ"].tolist(), int(g["num_species"]),
                                      cutoff_fn=str(g["cutoff_fn"]), row_capacity=256).to(dev)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    cell = torch.from_numpy(g["cell"]).to(dev) if "cell" in g else None
    pbc = torch.from_numpy(g["pbc"]) if "pbc" in g else None
    w = torch.from_numpy(g["cotangent"]).to(dev)
    xx = x.clone().requires_grad_(True)
    aev = aevc(sp, xx, cell, pbc)
    (vjp,) = torch.autograd.grad((aev * w).sum(), xx)
    torch.cuda.synchronize()
    aevc.last_neighbors().raise_on_overflow()
    err = np.abs(aev.detach().cpu().numpy() - g["aev"]).max()
    verr = np.abs(vjp.cpu().numpy() - g["aev_vjp"]).max()
    vmag = np.abs(g["aev_vjp"]).max()
    report(f"grid  {name:22s} max|aev err| = {err:.2e}   vjp err = {verr:.2e} (|vjp|max {vmag:.1f})")
    assert err < AEV_TOL * max(1.0, np.abs(g["aev"]).max())
    assert verr < 2e-5 * max(1.0, vmag)
    assert np.all(aev.detach().cpu().numpy()[g["species"] < 0] == 0)
    assert np.all(vjp.cpu().numpy()[g["species"] < 0] == 0)
    # engine level: virial, fixed point, shards
    eng = aevc.engine()
    sp32 = sp.to(torch.int32)
    pbc_t = None if pbc is None else tuple(bool(b) for b in pbc.tolist())
    nbrs = eng.neighbors(sp32, x, cell, pbc_t, mode="batch", row_cap=256)
    n = sp32.numel()
    wf = w.reshape(n, -1).contiguous()
    vir = torch.zeros((3, 3), dtype=torch.float64, device=dev)
    gc = eng.backward(sp32, nbrs, wf, virial=vir)
    assert float((gc.view_as(vjp) - vjp).abs().max()) < 1e-5 * max(1.0, vmag)
    # the virial of a translation-invariant scalar: W = sum_k g_k (x) r_k for an isolated system (no images)
    if cell is None:
        want = torch.einsum("nk,nb->kb", gc.double(), x.reshape(n, 3).double())
        assert float((vir - want).abs().max()) < 2e-4 * max(1.0, float(want.abs().max()))
    acc = torch.zeros((n, 3), dtype=torch.int64, device=dev)
    eng.backward(sp32, nbrs, wf, grad_coords=acc, fixed_point=True)
    from torchani_amd.engine import fixed_to_float
    assert float((fixed_to_float(acc) - gc).abs().max()) < 1e-5 * max(1.0, vmag)
    tot = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    for lo, hi in ((0, n // 3), (n // 3, n)):
        part = eng.neighbors(sp32, x, cell, pbc_t, lo=lo, hi=hi, mode="batch", row_cap=256)
        a_part = eng.forward(sp32, part, shard_rows=True)
        assert float((a_part - aev.detach().reshape(n, -1)[lo:hi]).abs().max()) < 1e-6
        eng.backward(sp32, part, wf[lo:hi].contiguous(), grad_coords=tot, shard_rows=True)
    assert float((tot - gc).abs().max()) < 1e-5 * max(1.0, vmag)
    # forward-mode derivative J t (anihip_aev_jvp on a general grid: the JVP instantiation of k_aev_fwd_gen) against the
    # reference's own (fixture), the adjoint identity <w, J t> = <J^T w, t> with the backward kernel, zero rows for padding
    from _util import fgrad_direction
    td = torch.from_numpy(fgrad_direction(g["species"]).astype(np.float32)).to(dev)
    jt = eng.jvp(sp32, nbrs, td)
    jscale = np.abs(g["aev_jvp"]).max()
    jerr = np.abs(jt.cpu().numpy().astype(np.float64).reshape(g["aev_jvp"].shape) - g["aev_jvp"]).max()
    report(f"grid  {name:22s} max|J t err| = {jerr:.2e} (max |J t| {jscale:.2f})")
    assert jerr < 2e-5 * max(1.0, jscale)
    lhs = (wf.double() * jt.double()).sum().item()
    rhs = (gc.double() * td.view(-1, 3).double()).sum().item()
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    assert torch.all(jt.view(sp.shape[0], sp.shape[1], -1)[sp < 0] == 0)
    # slab flags of the plain 32-column slabs: exactly the slabs that hold a non-zero entry or belong to a present block
    mask = torch.zeros(n, dtype=torch.int32, device=dev)
    a_m = eng.forward(sp32, nbrs, slab_mask=mask)
    assert torch.equal(a_m, aev.detach().reshape(n, -1).float()) or float((a_m - aev.detach().reshape(n, -1)).abs().max()) < 1e-6
    L = a_m.shape[1]
    pad_cols = (-L) % 32
    nz = torch.nn.functional.pad(a_m != 0, (0, pad_cols)).view(n, -1, 32).any(dim=2)   # [n, slabs]
    bits = ((mask.to(torch.int64).view(n, 1) & 0xFFFFFFFF) >> torch.arange(nz.shape[1], device=dev).view(1, -1)) & 1
    assert bool((bits.bool() | ~nz).all()), "a slab with a non-zero entry is not flagged"
    assert bool((bits.sum(dim=1)[sp32.view(-1) < 0] == 0).all())


def test_general_grids_random_sweep(dev):
    """Randomised grids (1 .. 32 radial shifts, angular grids 1x1 .. 16x16, 1 .. 7 species, both envelopes), dense and sparse
    clusters and periodic boxes: AEV rows and the gradient of a seeded functional from csrc/aev_generic.hip against the fp64
    oracle (which is pinned to the reference on general grids by tests/test_oracle_golden.py)."""
    from oracle import oracle as orc
    from oracle.oracle import Oracle

    from torchani_amd.aev import AEVComputer

    o64 = Oracle("f64")
    rs = np.random.RandomState(31)
    for trial in range(24):
        S = int(rs.randint(1, 8))
        nR, nA, nZ = int(rs.choice([1, 3, 8, 16, 17, 32])), int(rs.choice([1, 2, 5, 8, 16])), int(rs.choice([1, 3, 4, 9, 16]))
        if (nR, nA, nZ) in ((16, 8, 4), (16, 4, 8)):
            nR = 8   # (the published grids take the tuned kernels: tested elsewhere)
        Rcr, Rca = float(rs.uniform(3.5, 5.5)), float(rs.uniform(2.5, 3.5))
        EtaR, EtaA, Zeta = float(rs.uniform(4, 25)), float(rs.uniform(4, 15)), float(rs.choice([1.0, 8.0, 14.1, 32.0]))
        ShfR = np.linspace(0.8, Rcr - 0.3, nR).astype(np.float32)
        ShfA = np.linspace(0.8, Rca - 0.3, nA).astype(np.float32)
        ShfZ = ((np.arange(nZ) + 0.5) * np.pi / nZ).astype(np.float32)
        cut = "smooth" if rs.rand() < 0.4 else "cosine"
        C, A = (1, int(rs.choice([2, 30, 70]))) if rs.rand() < 0.6 else (3, 14)
        box = float(rs.uniform(3.0, 7.0)) if A < 50 else 5.5
        x = np.zeros((C, A, 3), dtype=np.float32)
        sp = rs.randint(0, S, (C, A)).astype(np.int32)
        for c in range(C):
            pts = []
            while len(pts) < A:
                q = rs.uniform(0, box, 3)
                if all(np.linalg.norm(q - w_) > 0.7 for w_ in pts):
                    pts.append(q)
            x[c] = np.asarray(pts, dtype=np.float32)
        if C > 1:
            sp[1, A - 3:] = -1
        cell = pbc = None
        if C == 1 and rs.rand() < 0.5:
            cell = (np.eye(3) * max(box, 2 * Rcr + 0.2)).astype(np.float32)
            cell[1, 0] = 0.7
            pbc = (True, True, bool(rs.rand() < 0.5))
        aevc = AEVComputer.from_constants(Rcr, Rca, EtaR, ShfR.tolist(), EtaA, Zeta, ShfA.tolist(), ShfZ.tolist(), S,
                                          cutoff_fn=cut, row_capacity=256).to(dev)
        p = orc.make_params(S, np.float32(Rcr), np.float32(Rca), EtaR, EtaA, Zeta, ShfR.tolist(), ShfA.tolist(), ShfZ.tolist(), cut)
        L = aevc.out_dim
        w = rs.uniform(-1, 1, (C, A, L)).astype(np.float32)
        ref_aev, ref_vjp = o64.aev(p, sp, x, cell, pbc, grad_aev=w.astype(np.float64))
        xx = torch.from_numpy(x).to(dev).requires_grad_(True)
        cell_t = None if cell is None else torch.from_numpy(cell).to(dev)
        pbc_t = None if pbc is None else torch.tensor(pbc)
        aev = aevc(torch.from_numpy(sp.astype(np.int64)).to(dev), xx, cell_t, pbc_t)
        (vjp,) = torch.autograd.grad((aev * torch.from_numpy(w).to(dev)).sum(), xx)
        torch.cuda.synchronize()
        aevc.last_neighbors().raise_on_overflow()
        scale_a, scale_v = max(1.0, np.abs(ref_aev).max()), max(1.0, np.abs(ref_vjp).max())
        err = np.abs(aev.detach().cpu().numpy() - ref_aev).max() / scale_a
        verr = np.abs(vjp.cpu().numpy() - ref_vjp).max() / scale_v
        assert err < AEV_TOL and verr < 3e-5, (trial, S, nR, nA, nZ, cut, C, A, pbc, err, verr)


def test_model_on_a_general_grid(dev):
    """A whole potential -- neighbor rows, general-grid AEVs, 3-member ensemble, analytic forces, self energies -- assembled
    from AEVComputer.from_constants with an 8 / 4x4 grid (192 columns) against the fp64 oracle: the networks take the
    dense layer-0 path (no slab structure), the AEV kernels the general ones."""
    from oracle import oracle as orc
    from oracle.oracle import Oracle

    from torchani_amd.aev import AEVComputer
    from torchani_amd.models import ANI
    from torchani_amd.nn import ANINetworks, Ensemble

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grid_r8_a4z4_batch.npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = ("H", "C", "N", "O")
    aevc = AEVComputer.from_constants(float(g["Rcr"]), float(g["Rca"]), float(g["EtaR"]), g["ShfR"].tolist(), float(g["EtaA"]),
                                      float(g["Zeta"]), g["ShfA"].tolist(), g["ShfZ"].tolist(), 4, row_capacity=256)
    hidden = {"H": (64, 48, 32), "C": (64, 32, 32), "N": (32, 32, 32), "O": (48, 32, 32)}
    torch.manual_seed(7)
    nets = Ensemble([ANINetworks.build(symbols, aevc.out_dim, hidden) for _ in range(3)])
    sae = [-0.5, -37.8, -54.6, -75.0]
    model = ANI(symbols, aevc, nets, sae, periodic_table_index=False).to(dev)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    out = model.energies_and_forces(sp, x)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    dims, flat = orc.pack_networks(sd, symbols, 3)
    p = orc.make_params(4, float(g["Rcr"]), float(g["Rca"]), float(g["EtaR"]), float(g["EtaA"]), float(g["Zeta"]),
                        g["ShfR"].tolist(), g["ShfA"].tolist(), g["ShfZ"].tolist(), "cosine")
    ref = Oracle("f64").energy_forces(p, g["species"], g["coords"], dims, flat, 3, sae=np.asarray(sae))
    assert np.abs(out.atomic_energies.cpu().numpy() - ref["atomic_energies"]).max() < E_ATOM_TOL
    assert np.abs(out.forces.cpu().numpy() - ref["forces"]).max() < F_TOL
    assert np.abs(out.energies.cpu().numpy() - ref["energies"]).max() < 1e-5
    # the same through autograd
    from torchani_amd.grad import energies_and_forces
    e, f = energies_and_forces(model, sp, x)
    assert np.abs(f.cpu().numpy() - ref["forces"]).max() < F_TOL


NBR_CASES = ["simple2_ani2x", "rand_batch_ani2x", "water_pbc_ani2x", "triclinic_pbc_ani2x", "small_ani2x",
             "ch4_ani1x"]


@pytest.mark.parametrize("name", NBR_CASES)
def test_compute_from_external_half_list(dev, name):
    """AEVComputer.compute_from_neighbors on the REFERENCE's own half neighbor list (tests/golden/nbrs_*.npz,
    written by gen_golden_nbrs.py): same AEVs and coordinate gradients as the golden fixture, and the same
    neighbor rows as the engine's own builder (up to the order inside a species group)."""
    from torchani_amd.aev import AEVComputer
    from torchani_amd.tuples import Neighbors
    from torchani_amd.weights import arch_spec

    g = load_golden(name)
    nb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"nbrs_{name}.npz"))
    consts = arch_spec(g["kind"])[1]._replace(cutoff_fn=g["cutoff_fn"])
    sp, x, cell, pbc = to_dev(g, dev)
    C, A = g["species"].shape
    idx = torch.from_numpy(nb["indices"]).to(dev)
    diff = torch.from_numpy(nb["diff_vectors"]).to(dev)
    aevc = AEVComputer(consts, row_capacity=256).to(dev)
    w = torch.from_numpy(np.random.RandomState(g["seed"] + 1000).uniform(-1.0, 1.0, (C, A, consts.out_dim))
                         .astype(np.float32)).to(dev)
    xx = x.clone().requires_grad_(True)
    aev = aevc.compute_from_neighbors(sp, xx, Neighbors(idx, diff.norm(dim=-1), diff))
    (vjp,) = torch.autograd.grad((aev * w).sum(), xx)
    torch.cuda.synchronize()
    aevc.last_neighbors().raise_on_overflow()
    got = aev.detach().cpu().numpy().reshape(C * A, -1)
    err = np.abs(got[g["aev_rows"]] - g["aev"]).max()
    verr = np.abs(vjp.cpu().numpy() - g["aev_vjp"]).max()
    report(f"half  {name:22s} pairs={idx.shape[1]:5d} max|aev err| = {err:.2e}   vjp err = {verr:.2e}")
    assert err < AEV_TOL
    assert verr < 2e-5 * max(1.0, np.abs(g["aev_vjp"]).max())
    assert np.all(got[g["species"].reshape(-1) < 0] == 0)
    # rows: same neighbor multiset per atom as the engine's own list
    meta_h, ent_h = unpack_rows(aevc.last_neighbors(), C * A)
    pbc_t = None if pbc is None else torch.tensor(pbc)
    aevc2 = AEVComputer(consts, neighborlist="all_pairs", row_capacity=256).to(dev)
    aevc2(sp, x, cell, pbc_t)
    meta_o, ent_o = unpack_rows(aevc2.last_neighbors(), C * A)
    assert np.array_equal(meta_h[:, 1:], meta_o[:, 1:])   # counts per group and species
    for i in range(C * A):
        n = int(meta_h[i, 1] & 0xFFFF) + int(meta_h[i, 1] >> 16)
        a = ent_h[meta_h[i, 0]:meta_h[i, 0] + n]
        b = ent_o[meta_o[i, 0]:meta_o[i, 0] + n]
        def canon(rows):   # order by (packed neighbor, displacement), compare displacements with a tolerance
            k = np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0], rows[:, 3].view(np.uint32)))
            return rows[k]
        a, b = canon(a), canon(b)
        assert np.array_equal(a[:, 3].view(np.uint32), b[:, 3].view(np.uint32)), f"atom {i}"
        assert np.abs(a[:, :3] - b[:, :3]).max(initial=0.0) < 5e-6, f"atom {i}"


@pytest.mark.parametrize("name", ["small_ani2x", "ch4_ani1x"])
def test_compute_from_full_neighbor_list(dev, oracle64, name):
    """LAMMPS-style full list (aev/_computer.py:420-438, tests/test_cuaev.py:740-790): built here from the
    reference's half list of a non-periodic case (both directions, plus some pairs beyond the cutoff and a
    self pair that must be screened out); atoms left out of ilist get zero rows."""
    from torchani_amd.aev import AEVComputer
    from torchani_amd.weights import arch_spec

    g = load_golden(name)
    nb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"nbrs_{name}.npz"))
    consts = arch_spec(g["kind"])[1]._replace(cutoff_fn=g["cutoff_fn"])
    sp, x, _, _ = to_dev(g, dev)
    n = sp.numel()
    i0, i1 = nb["indices"]
    adj = [[] for _ in range(n)]
    for a, b in zip(i0, i1):
        adj[a].append(int(b))
        adj[b].append(int(a))
    rs = np.random.RandomState(0)
    for a in range(n):   # skin: extra candidates incl. the atom itself; the kernel screens by distance
        adj[a] += [int(v) for v in rs.randint(0, n, 3)] + [a]
        adj[a] = list(dict.fromkeys(adj[a]))
    listed = [a for a in range(n) if a % 7 != 3]   # leave some atoms out
    ilist = torch.tensor(listed, dtype=torch.int32, device=dev)
    numneigh = torch.tensor([len(adj[a]) for a in listed], dtype=torch.int32, device=dev)
    jlist = torch.tensor([j for a in listed for j in adj[a]], dtype=torch.int32, device=dev)
    aevc = AEVComputer(consts, row_capacity=256).to(dev)
    xx = x.clone().requires_grad_(True)
    aev = aevc.compute_from_full_nbrlist(sp, xx, ilist, jlist, numneigh)
    torch.cuda.synchronize()
    aevc.last_neighbors().raise_on_overflow()
    got = aev.detach().cpu().numpy().reshape(n, -1)
    ref = np.zeros_like(got)
    ref[g["aev_rows"]] = g["aev"]
    mask = np.zeros(n, dtype=bool)
    mask[listed] = True
    rows = np.intersect1d(g["aev_rows"], np.asarray(listed))
    assert np.abs(got[rows] - ref[rows]).max() < AEV_TOL
    assert np.all(got[~mask] == 0)
    # the backward through a full list: rows of unlisted atoms do not exist (not even as somebody's partner), so the
    # radial gather-by-symmetry must be off; VJP with seeded cotangents == the oracle's with the cotangents of the
    # unlisted atoms zeroed
    w = np.random.RandomState(7).uniform(-1.0, 1.0, got.shape)
    (gx,) = torch.autograd.grad((aev.view(n, -1) * torch.from_numpy(w.astype(np.float32)).to(dev)).sum(), xx)
    p = oracle_params(g["kind"], g["cutoff_fn"])
    _, gc_ref = oracle64.aev(p, g["species"], g["coords"].astype(np.float64), grad_aev=w * mask[:, None])
    scale = max(1.0, np.abs(gc_ref).max())
    assert np.abs(gx.cpu().numpy().reshape(gc_ref.shape) - gc_ref).max() < 2e-5 * scale
    # everything listed: the golden VJP itself
    all_i = torch.arange(n, dtype=torch.int32, device=dev)
    nn_all = torch.tensor([len(adj[a]) for a in range(n)], dtype=torch.int32, device=dev)
    jl_all = torch.tensor([j for a in range(n) for j in adj[a]], dtype=torch.int32, device=dev)
    x2 = x.clone().requires_grad_(True)
    aev2 = aevc.compute_from_full_nbrlist(sp, x2, all_i, jl_all, nn_all)
    wg = np.random.RandomState(g["seed"] + 1000).uniform(-1.0, 1.0, tuple(g["species"].shape) + (got.shape[1],))
    if aev2.shape[0] == g["aev_vjp"].shape[0]:
        (gx2,) = torch.autograd.grad((aev2 * torch.from_numpy(wg.astype(np.float32)).to(dev)).sum(), x2)
        assert np.abs(gx2.cpu().numpy() - g["aev_vjp"]).max() < 2e-5 * max(1.0, np.abs(g["aev_vjp"]).max())


@pytest.mark.parametrize("name", ["rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x"])
def test_model_from_external_neighbors(dev, name):
    """ANI.compute_from_neighbors / compute_from_external_neighbors (arch.py:171-206,354-381): energies and
    autograd forces from the reference's half list equal the golden values; a Verlet-skin style list with
    extra, longer pairs gives the same result."""
    g = load_golden(name)
    nb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"nbrs_{name}.npz"))
    sp, x, cell, pbc = to_dev(g, dev)
    model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"], row_capacity=256)
    idx = torch.from_numpy(nb["indices"]).to(dev)
    diff = torch.from_numpy(nb["diff_vectors"]).to(dev)
    xx = x.clone().requires_grad_(True)
    res = model.compute_from_neighbors(sp, xx, (idx, diff.norm(dim=-1), diff))
    assert res.scalars is None   # EnergiesScalars like the reference (arch.py:353-381)
    e = res.energies
    (gx,) = torch.autograd.grad(e.sum(), xx)
    assert np.abs(e.detach().double().cpu().numpy() - g["energies"]).max() < 2e-6 * np.abs(g["energies"]).max()
    assert np.abs(-gx.cpu().numpy() - g["forces"]).max() < F_TOL
    # external list: cartesian shifts = diff - (r0 - r1); add far pairs that the screening must drop
    flat = x.reshape(-1, 3)
    shifts = diff - (flat[idx[0]] - flat[idx[1]])
    n = flat.shape[0]
    extra = torch.tensor([[0], [n - 1]], device=dev)
    far = torch.tensor([[40.0, 0.0, 0.0]], device=dev) - (flat[0] - flat[n - 1]).unsqueeze(0)
    idx2, shifts2 = torch.cat([idx, extra], 1), torch.cat([shifts, far], 0)
    e2 = model.compute_from_external_neighbors(sp, x, idx2, shifts2).energies
    assert torch.allclose(e2, e.detach(), rtol=0, atol=1e-5)


@pytest.mark.parametrize("precision", ["f16x3", "f16x3-unfused", "f16x3-bigtile", "fp32"])
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_mlp_ensemble(dev, oracle64, name, precision, monkeypatch):
    """Networks alone: reference-exact AEVs in, per-atom energies and d/d aev out, for both GEMM
    arithmetics (exact fp32 MFMA and the split-fp16 three-product MFMA)."""
    g = load_golden(name)
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    aev = oracle64.aev(p, g["species"], g["coords"].astype(np.float64), g["cell"], g["pbc"])
    ae, ga, me = oracle64.mlp(g["species"], aev, dims, flat, n_members=8, want_members=True)
    model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"])
    if precision == "f16x3-unfused":  # layer-by-layer f16x3 GEMMs instead of the fused network kernel
        monkeypatch.setattr(PackedNetworks, "default_flags", _lib.MLP_FLAG_NO_FUSED)
    if precision == "f16x3-bigtile":  # force the 256x256-tile layer-0 GEMM that large systems use
        monkeypatch.setattr(PackedNetworks, "default_flags", _lib.MLP_FLAG_BIG_TILES)
    model.neural_networks.mlp_precision = precision.split("-")[0]
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    a32 = torch.from_numpy(aev.astype(np.float32)).to(dev).requires_grad_(True)
    e = model.neural_networks(sp, a32, atomic=True)
    (gr,) = torch.autograd.grad(e.sum(), a32)
    em = model.neural_networks(sp, a32.detach(), atomic=True, ensemble_values=True)
    torch.cuda.synchronize()
    model.neural_networks.mlp_precision = None
    C, A = g["species"].shape
    e_err = np.abs(e.detach().cpu().numpy() - ae.reshape(C, A)).max()
    g_err = np.abs(gr.cpu().numpy().reshape(C * A, -1) - ga).max()
    m_err = np.abs(em.cpu().numpy() - me.reshape(8, C, A)).max()
    report(f"mlp   {name:22s} {precision:13s} max|e_atom err| = {e_err:.2e}  |d e/d aev err| = {g_err:.2e} "
           f"(max {np.abs(ga).max():.2e})  members {m_err:.2e}")
    assert e_err < E_ATOM_TOL and m_err < E_ATOM_TOL
    # regression guard well inside the parity gate: every arithmetic variant sits at the fp32 round-off level
    # (<= 8e-8 Ha per atom, 3.3e-7 per member on these cases); a lost fp16 "lo" plane shows up as ~1e-6
    assert e_err < 3e-7 and m_err < 1.5e-6
    assert g_err < 1e-6 + 1e-5 * np.abs(ga).max()
    pad = g["species"] < 0
    assert np.all(e.detach().cpu().numpy()[pad] == 0) and np.all(gr.cpu().numpy()[pad] == 0)


def test_force_training_on_a_general_grid(dev, oracle64):
    """The reference's force-training recipe (forces = -autograd.grad(E, coords, create_graph=True); loss(forces).backward(),
    tools/training-aev-benchmark.py:136-150) on a from_constants grid: until round 4 the general kernels had no
    forward-mode derivative and this raised.  Parameter gradients against the oracle's second-order chain (pinned to the
    reference on the same grids: test_oracle_golden.test_oracle_on_general_grids)."""
    from _util import fgrad_direction
    from oracle import oracle as orc

    from torchani_amd.aev import AEVComputer
    from torchani_amd.models import ANI
    from torchani_amd.nn import ANINetworks, Ensemble

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grid_r8_a4z4_batch.npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = ("H", "C", "N", "O")
    aevc = AEVComputer.from_constants(float(g["Rcr"]), float(g["Rca"]), float(g["EtaR"]), g["ShfR"].tolist(), float(g["EtaA"]),
                                      float(g["Zeta"]), g["ShfA"].tolist(), g["ShfZ"].tolist(), 4, row_capacity=256)
    hidden = {"H": (64, 48, 32), "C": (64, 32, 32), "N": (32, 32, 32), "O": (48, 32, 32)}
    torch.manual_seed(11)
    M = 2
    nets = Ensemble([ANINetworks.build(symbols, aevc.out_dim, hidden) for _ in range(M)])
    model = ANI(symbols, aevc, nets, [0.0, 0.0, 0.0, 0.0], periodic_table_index=False).to(dev)
    nets.requires_grad_(True)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev).requires_grad_(True)
    t_np = fgrad_direction(g["species"])
    t = torch.from_numpy(t_np.astype(np.float32)).to(dev)
    aev = model.aev_computer(sp, x)
    e = nets(sp, aev).sum()
    (gx,) = torch.autograd.grad(e, x, create_graph=True)
    loss = -(gx * t).sum()                      # = sum_k t_k . F_k
    loss.backward()
    torch.cuda.synchronize()
    # the oracle: S = sum_i v_i . d e_i / d aev_i with v = -J t, differentiated with respect to every parameter
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    dims, flat = orc.pack_networks(sd, symbols, M)
    p = orc.make_params(4, float(g["Rcr"]), float(g["Rca"]), float(g["EtaR"]), float(g["EtaA"]), float(g["Zeta"]),
                        g["ShfR"].tolist(), g["ShfA"].tolist(), g["ShfZ"].tolist(), "cosine")
    aev_ref, jt_ref = oracle64.aev_jvp(p, g["species"], g["coords"].astype(np.float64), t_np)
    val_ref, ref = oracle64.mlp_tangent_weight_grads(g["species"], aev_ref, -jt_ref, dims, flat, n_members=M)
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().cpu().numpy()
             for k, v in model.named_parameters()}
    _, got = orc.pack_networks({k: grads.get(k, v) for k, v in sd.items()}, symbols, M)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    report(f"ftrain general grid 8 / 4x4: |loss err| = {abs(float(loss.detach()) - val_ref):.2e}  max|d(t.F)/dw err| = {err:.2e} (max {scale:.2e})")
    assert abs(float(loss.detach()) - val_ref) < 1e-5 * max(1.0, abs(val_ref))
    assert scale > 0 and err < 5e-5 * scale


def test_general_grid_slab_flags_through_the_networks(dev, oracle64):
    """A general grid whose networks run through the fused kernel (three hidden layers, widths >= 128 in the middle so that
    the layer-0 backward can run inside it): the flags of the plain 32-column slabs from k_aev_fwd_gen make layer 0 skip the
    slabs of absent species -- same energies as without flags, forces against the oracle, also with phase 5 forced."""
    from bench import water_box
    from oracle import oracle as orc

    from torchani_amd.aev import AEVComputer
    from torchani_amd.models import ANI
    from torchani_amd.nn import ANINetworks, Ensemble

    symbols = ("H", "C", "N", "O")
    aevc = AEVComputer.from_constants(5.1, 3.5, 19.7, np.linspace(0.8, 4.8, 12).tolist(), 12.5, 14.1,
                                      np.linspace(0.8, 3.1, 6).tolist(), ((np.arange(4) + 0.5) * np.pi / 4).tolist(), 4,
                                      row_capacity=192, neighborlist="cell")   # 48 + 10 x 24 = 288 columns = 9 slabs
    torch.manual_seed(13)
    M = 2
    nets = Ensemble([ANINetworks.build(symbols, aevc.out_dim, {s: (160, 128, 96) for s in symbols}) for _ in range(M)])
    model = ANI(symbols, aevc, nets, [0.0, 0.0, 0.0, 0.0], periodic_table_index=False).to(dev)
    sp_np, x_np, cell_np = water_box(10)   # 3000 atoms, species 0 and 3
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    eng, packed = model.aev_computer.engine(), model.neural_networks._pack(dev)
    assert model._plain_slabs(eng, packed)
    out = model.energies_and_forces(sp, x, cell, pbc, check_overflow=True)
    old = PackedNetworks.default_flags
    try:
        PackedNetworks.default_flags = _lib.MLP_FLAG_NO_SLAB_MASK
        dense = model.energies_and_forces(sp, x, cell, pbc)
        PackedNetworks.default_flags = _lib.MLP_FLAG_FUSED_L0B
        inside = model.energies_and_forces(sp, x, cell, pbc)
    finally:
        PackedNetworks.default_flags = old
    assert float((out.atomic_energies - dense.atomic_energies).abs().max()) < 2e-7   # (other k steps share a tile scale)
    assert float((out.forces - dense.forces).abs().max()) < 2e-6 and float((inside.forces - dense.forces).abs().max()) < 2e-6
    # and the flags did skip slabs: water under H C N O has no C / N neighbors
    nbrs = model.aev_computer.last_neighbors()
    mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
    eng.forward(sp.to(torch.int32), nbrs, slab_mask=mask)
    assert 0 < bin(int(mask[0].item()) & 0xFFFFFFFF).count("1") < 9
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    dims, flat = orc.pack_networks(sd, symbols, M)
    p = orc.make_params(4, 5.1, 3.5, 19.7, 12.5, 14.1, np.linspace(0.8, 4.8, 12).tolist(), np.linspace(0.8, 3.1, 6).tolist(),
                        ((np.arange(4) + 0.5) * np.pi / 4).tolist(), "cosine")
    ref = oracle64.energy_forces(p, sp_np, x_np.astype(np.float64), dims, flat, M, sae=np.zeros(4), cell=cell_np,
                                 pbc=pbc, cell_list=True)
    ea = np.abs(out.atomic_energies.cpu().numpy() - ref["atomic_energies"]).max()
    fe = np.abs(out.forces.cpu().numpy() - ref["forces"]).max()
    report(f"grid  12 / 6x4 water 3000 atoms, flagged slabs: max|e_atom err| = {ea:.2e}  |F err| = {fe:.2e}")
    assert ea <= E_ATOM_REG and fe <= F_REG


def test_padding_atoms_in_a_large_system(dev):
    """Padding (species -1) in a system large enough for the chunked species bucketing (> 16 384 atoms; the padding rows are
    zeroed by k_sp_scatter since round 4): the padded atoms get zero energies and forces, the others what the same system
    WITHOUT the padded atoms gives."""
    from bench import water_box

    sp_np, x_np, cell_np = water_box(18)   # 17 496 atoms
    n = sp_np.shape[1]
    keep = (np.arange(n) % 7) != 3
    sp_pad = sp_np.copy()
    sp_pad[0, ~keep] = -1
    model = get_model("ani2x", 2, dev, neighborlist="cell", row_capacity=192)
    cell = torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    a = model.energies_and_forces(torch.from_numpy(sp_pad).to(dev), torch.from_numpy(x_np).to(dev), cell, pbc, check_overflow=True)
    b = model.energies_and_forces(torch.from_numpy(np.ascontiguousarray(sp_np[:, keep])).to(dev),
                                  torch.from_numpy(np.ascontiguousarray(x_np[:, keep])).to(dev), cell, pbc, check_overflow=True)
    kt = torch.from_numpy(keep).to(dev)
    assert float(a.atomic_energies[0, ~kt].abs().max()) == 0.0 and float(a.forces[0, ~kt].abs().max()) == 0.0
    assert float((a.atomic_energies[0, kt] - b.atomic_energies[0]).abs().max()) < 2e-7
    assert float((a.forces[0, kt] - b.forces[0]).abs().max()) < 2e-6
    assert abs(float(a.energies - b.energies)) < 1e-7 * n


def test_steady_state_step_does_not_synchronize(dev):
    """zero host synchronisations on the steady-state path.  After the caches are warm (species validity,
    species relabelling, tile hint, locality test: each reads the device ONCE per species tensor), an energies_and_forces
    call with check_overflow=False queues its kernels and returns -- torch's sync debug mode turns any synchronising call
    into an error.  (Round 4 found the last one: SpeciesConverter's validity check, a .max() per call like the reference's.)"""
    from bench import water_box

    sp_np, x_np, cell_np = water_box(29)   # 73 167 atoms: the large-system path (kept AEV rows, phase 5)
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    model = get_model("ani2x", 0, dev, neighborlist="cell")
    pbc = (True, True, True)
    for k in range(3):
        ref = model.energies_and_forces(sp, x + 0.001 * k, cell, pbc, check_overflow=False)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = model.energies_and_forces(sp, x + 0.002, cell, pbc, check_overflow=False)
        small = model.energies_and_forces(sp, x + 0.003, cell, pbc, check_overflow=False)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert float((out.forces - ref.forces).abs().max()) < 1e-5 and torch.isfinite(small.energies).all()
    model.aev_computer.last_neighbors().raise_on_overflow()


def test_aev_rows_updated_in_place(dev):
    """AevEngine.forward_update (anihip_aev_forward_update): the rows kept by the engine and updated in place are, after
    every call, bit for bit the rows anihip_aev_forward writes into a fresh buffer -- whatever happened to the system in
    between: other elements (other slabs flagged: the old ones must be zeroed, the new ones written), atoms turned into
    padding, atoms that lost all their neighbors, the first use."""
    from bench import water_box

    sp_np, x_np, cell_np = water_box(10)   # 3000 atoms
    x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    model = get_model("ani2x", 3, dev, neighborlist="cell")
    eng = model.aev_computer.engine()
    rs = np.random.RandomState(1)
    n = sp_np.size
    base = torch.from_numpy(sp_np).to(dev).to(torch.int32)
    far = x.clone()
    far[0, ::7] += torch.tensor([0.0, 0.0, 1000.0], device=dev)   # (open boundaries below: these atoms are alone)
    steps = [
        (base, x, (True, True, True)),
        (torch.where(base == 0, 1, 2).to(torch.int32), x, (True, True, True)),
        (torch.from_numpy(rs.choice([0, 1, 2, 3, 4, 5, 6], size=(1, n))).to(dev).to(torch.int32), x, (True, True, True)),
        (torch.where(torch.from_numpy(rs.rand(1, n) < 0.3).to(dev), -1, base).to(torch.int32), x, (True, True, True)),
        (base, far, (False, False, False)),
        (base, x, (True, True, True)),
    ]
    for k, (sp, xx, pbc) in enumerate(steps):
        nbrs = eng.neighbors(sp, xx, cell, pbc, mode="cell", row_cap=160)
        kept, kept_mask = eng.forward_update(sp, nbrs, shard_rows=False)
        mask = torch.zeros(n, dtype=torch.int32, device=dev)
        fresh = eng.forward(sp, nbrs, slab_mask=mask)
        assert torch.equal(kept, fresh), (k, float((kept - fresh).abs().max()))
        assert torch.equal(kept_mask, mask), k
        if k == 0:
            first = kept.data_ptr()
        assert kept.data_ptr() == first   # (the same buffers every time)
    # and through the model: energies_and_forces with kept rows == with fresh rows, bit for bit, call after call
    sp64 = base.to(torch.int64)
    model.keep_aev_rows = False
    ref = [model.energies_and_forces(sp64, xx, cell, (True, True, True)) for xx in (x, x + 0.05, x)]
    model.keep_aev_rows = True
    got = [model.energies_and_forces(sp64, xx, cell, (True, True, True)) for xx in (x, x + 0.05, x)]
    for a, b in zip(ref, got):
        assert torch.equal(a.atomic_energies, b.atomic_energies) and torch.equal(a.energies, b.energies)
        assert float((a.forces - b.forces).abs().max()) < 2e-6   # (float atomics in the AEV backward)


def test_general_grid_on_a_large_system(dev):
    """A from_constants grid (8 radial / 4 x 4 angular terms) on a 17 496-atom H / O box: large enough for the species
    relabelling of models.ANI._engine_species (water under H C N O would be relabelled (0, 3, 1, 2)), which only applies to
    the 16 / 32-column ANI layout -- a general grid must be left alone (it used to raise in the packer).  The fused path
    must agree with the autograd path, which never relabels."""
    from bench import water_box
    from torchani_amd.aev import AEVComputer
    from torchani_amd.grad import energies_and_forces
    from torchani_amd.models import ANI
    from torchani_amd.nn import ANINetworks, Ensemble

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grid_r8_a4z4_batch.npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = ("H", "C", "N", "O")
    aevc = AEVComputer.from_constants(float(g["Rcr"]), float(g["Rca"]), float(g["EtaR"]), g["ShfR"].tolist(), float(g["EtaA"]),
                                      float(g["Zeta"]), g["ShfA"].tolist(), g["ShfZ"].tolist(), 4, row_capacity=256,
                                      neighborlist="cell")
    torch.manual_seed(9)
    nets = Ensemble([ANINetworks.build(symbols, aevc.out_dim, {"H": (64, 48, 32), "C": (64, 32, 32), "N": (32, 32, 32),
                                                               "O": (48, 32, 32)}) for _ in range(2)])
    model = ANI(symbols, aevc, nets, [-0.5, -37.8, -54.6, -75.0], periodic_table_index=False).to(dev)
    sp_np, x_np, cell_np = water_box(18)   # species 0 (H) and 3 (O)
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    assert model._engine_species(sp.to(torch.int32))[1] is None
    out = model.energies_and_forces(sp, x, cell, pbc, check_overflow=True)
    e, f = energies_and_forces(model, sp, x, cell, torch.tensor(pbc))
    assert abs(float(out.energies - e)) < 0.05   # (the autograd path returns the total in fp32: one ulp of 4.4e5 Ha is 0.03)
    assert float((out.forces - f).abs().max()) < 5e-6 * max(1.0, float(f.abs().max()))


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_energies_and_forces_fused(dev, name):
    g = load_golden(name)
    sp, x, cell, pbc = to_dev(g, dev)
    for mode in modes_for(g):
        model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"], neighborlist=mode, row_capacity=256)
        out = model.energies_and_forces(sp, x, cell, pbc, check_overflow=True)
        torch.cuda.synchronize()
        ea = np.abs(out.atomic_energies.cpu().numpy() - g["atomic_energies"]).max()
        et = np.abs(out.energies.cpu().numpy() - g["energies"]).max()
        fe = np.abs(out.forces.cpu().numpy() - g["forces"]).max()
        n_real = int((g["species"] >= 0).sum())
        report(f"e+f   {name:22s} {mode:5s} atoms={n_real:4d} max|e_atom err| = {ea:.2e}  |E_total err| = {et:.2e}"
               f"  |F err| = {fe:.2e}")
        assert ea < E_ATOM_TOL
        assert fe < F_TOL
        assert ea <= E_ATOM_REG and fe <= F_REG, "regression gate (module header)"
        # totals are accumulated in fp64 from fp32 per-atom energies: error grows at most like n * eps
        assert et < E_ATOM_TOL * max(1.0, np.sqrt(n_real))
        assert np.all(out.forces.cpu().numpy()[g["species"] < 0] == 0)


@pytest.mark.parametrize("name", ["cfg2_xyz13_28_ani2x", "rand_batch_ani2x", "small_ani2x"])
def test_small_inputs_are_reproducible_and_leave_padding_rows_zero(dev, name, monkeypatch):
    """Below 16384 atoms the bucketing / tile table / padding rows are ONE launch (k_small_prep, a stable counting sort) and
    the layer-0 backward is the 8-wave 128 x 128 kernel: with fixed-point forces two evaluations are bit-identical, and the
    rows of padding atoms come out zero.  (Rounds 2-4 compared this path bit for bit with the launch-by-launch variants it
    replaced; those switches were retired in round 5.)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    with np.load(path) as z:
        sp = torch.from_numpy(z["species"].astype(np.int64)).to(dev)
        x = torch.from_numpy(z["coords"].astype(np.float32)).to(dev)
    model = get_model("ani2x", 3, dev, neighborlist="batch", row_capacity=256)
    monkeypatch.setattr(model, "auto_graph_atoms", 0)
    monkeypatch.setattr(model, "deterministic_forces", True)
    a = model.energies_and_forces(sp, x, check_overflow=True)
    a = (a.energies.clone(), a.forces.clone(), a.atomic_energies.clone())
    b = model.energies_and_forces(sp, x, check_overflow=True)
    for u, v in zip(a, (b.energies, b.forces, b.atomic_energies)):
        assert torch.equal(u, v)
    assert (a[1][sp < 0] == 0).all() and (a[2][sp < 0] == 0).all()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_energies_and_forces_slab_masks(dev, name, monkeypatch):
    """The large-system configuration of the fused path on the golden cases: 256x256-tile layer-0 GEMMs
    that skip the AEV slabs of absent neighbor species (per-atom slab masks from the AEV kernel)."""
    monkeypatch.setattr(PackedNetworks, "default_flags", _lib.MLP_FLAG_BIG_TILES)
    g = load_golden(name)
    sp, x, cell, pbc = to_dev(g, dev)
    model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"], neighborlist=modes_for(g)[-1], row_capacity=256)
    out = model.energies_and_forces(sp, x, cell, pbc, check_overflow=True)
    torch.cuda.synchronize()
    ea = np.abs(out.atomic_energies.cpu().numpy() - g["atomic_energies"]).max()
    fe = np.abs(out.forces.cpu().numpy() - g["forces"]).max()
    report(f"slab  {name:22s} max|e_atom err| = {ea:.2e}  |F err| = {fe:.2e}")
    assert ea < E_ATOM_TOL and fe < F_TOL
    assert ea <= E_ATOM_REG and fe <= F_REG, "regression gate (module header)"
    assert np.all(out.forces.cpu().numpy()[g["species"] < 0] == 0)


@pytest.mark.parametrize("name", ["small_ani2x", "water_pbc_ani2x", "ch4_ani1x", "1hz5_ani2x"])
def test_slab_masks_flag_exactly_the_nonzero_blocks(dev, name, monkeypatch):
    """slab_mask[i] bit j set <=> atom i has a neighbor (pair) contributing to slab j; masked and dense
    layer-0 GEMMs give the same energies, and the same gradients inside the flagged slabs."""
    from torchani_amd.weights import arch_spec

    monkeypatch.setattr(PackedNetworks, "default_flags", _lib.MLP_FLAG_BIG_TILES)
    g = load_golden(name)
    sp, x, cell, pbc = to_dev(g, dev)
    model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"])
    consts = arch_spec(g["kind"])[1]._replace(cutoff_fn=g["cutoff_fn"])
    eng = model.aev_computer.engine()
    sp32 = sp.to(torch.int32).contiguous().view(-1)
    n = sp32.numel()
    nbrs = eng.neighbors(sp32.view(sp.shape), x, cell, pbc, mode=modes_for(g)[-1], row_cap=256)
    mask = torch.full((n,), -1, dtype=torch.int32, device=dev)
    aev = eng.forward(sp32, nbrs, slab_mask=mask)
    S = eng.params.num_species
    rs, R = (S + 1) // 2, 16 * S
    a = aev.cpu().numpy()
    mk = mask.cpu().numpy().astype(np.uint32)
    # expected flags from the AEV rows themselves (a block with a contributing neighbor is > 0 somewhere
    # unless every term underflows; the converse -- unflagged => all zero -- must hold exactly)
    for j in range(eng.n_slabs):
        if j < rs:
            blk = a[:, 32 * j:min(32 * j + 32, R)]
        else:
            blk = a[:, R + 32 * (j - rs):R + 32 * (j - rs) + 32]
        flagged = (mk >> j) & 1 == 1
        assert np.all(blk[~flagged] == 0), f"slab {j}: non-zero AEV entries outside the mask"
    assert np.all(mk[g["species"].reshape(-1) < 0] == 0)
    packed = model.neural_networks._pack(dev)
    assert packed.radial_len == R
    e0, g0, _ = packed.forward_backward(sp32, aev)
    g1 = torch.full_like(g0, float("nan"))
    e1, g1, _ = packed.forward_backward(sp32, aev, slab_mask=mask, grad_aev=g1)
    torch.cuda.synchronize()
    assert torch.allclose(e0, e1, atol=2e-7, rtol=0)
    g0n, g1n = g0.cpu().numpy(), g1.cpu().numpy()
    real = g["species"].reshape(-1) >= 0
    for j in range(eng.n_slabs):
        cols = slice(32 * j, min(32 * j + 32, R)) if j < rs else slice(R + 32 * (j - rs), R + 32 * (j - rs) + 32)
        flagged = ((mk >> j) & 1 == 1) & real
        d = np.abs(g0n[flagged, cols] - g1n[flagged, cols])
        assert d.size == 0 or d.max() <= 1e-6 * max(1.0, np.abs(g0n).max())
    # forces through the masked gradient
    f0 = eng.backward(sp32, nbrs, g0)
    f1 = eng.backward(sp32, nbrs, g1)
    assert torch.isfinite(f1).all() and torch.allclose(f0, f1, atol=2e-6)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["rand_batch_ani2x", "small_ani2x", "water_pbc_ani2x", "1hz5_ani2x"])
def test_shards_add_up_to_the_whole(dev, name, world, monkeypatch):
    """The multi-GPU decomposition on one device: every rank's shard of the central atoms evaluated alone
    (what a rank does before the all-reduces of models.ANI.energies_and_forces); the partial energies and
    forces must add up to the golden result.  Run in the large-system configuration (slab masks)."""
    monkeypatch.setattr(PackedNetworks, "default_flags", _lib.MLP_FLAG_BIG_TILES)
    g = load_golden(name)
    sp, x, cell, pbc = to_dev(g, dev)
    model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"], neighborlist=modes_for(g)[-1], row_capacity=256)
    e = torch.zeros(sp.shape[0], dtype=torch.float64, device=dev)
    f = torch.zeros_like(x)
    ae = torch.zeros(sp.shape, dtype=torch.float32, device=dev)
    for rank in range(world):
        out = model.energies_and_forces(sp, x, cell, pbc, shard=(rank, world), check_overflow=True)
        e += out.energies
        f += out.forces
        ae += out.atomic_energies
    torch.cuda.synchronize()
    assert np.abs(ae.cpu().numpy() - g["atomic_energies"]).max() < E_ATOM_TOL
    assert np.abs(f.cpu().numpy() - g["forces"]).max() < F_TOL
    assert np.abs(ae.cpu().numpy() - g["atomic_energies"]).max() <= E_ATOM_REG and np.abs(f.cpu().numpy() - g["forces"]).max() <= F_REG
    n_real = int((g["species"] >= 0).sum())
    assert np.abs(e.cpu().numpy() - g["energies"]).max() < E_ATOM_TOL * max(1.0, np.sqrt(n_real))


def test_large_shards_pick_their_own_launch_scheme(dev):
    """Spatial shards of a system whose ranks own >= 24 000 atoms each: a rank prices the two launch schemes of the network stage
    for the atoms IT owns (models.ANI._per_species_launches_pay: 40 500 water atoms -> one launch per species, compile-time
    widths) -- central range in the middle of the local system, halo rows around it.  The two ranks' energies and forces add
    up to those of the whole 81 000-atom box evaluated at once."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from bench import water_box

    sp_np, x_np, cell_np = water_box(30)   # 81 000 atoms
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    model = get_model("ani2x", 0, dev, neighborlist="cell", row_capacity=160)
    whole = model.energies_and_forces(sp, x, cell, pbc)
    e = torch.zeros(1, dtype=torch.float64, device=dev)
    f = torch.zeros_like(x)
    for rank in range(2):
        part = model.energies_and_forces(sp, x, cell, pbc, shard=(rank, 2))
        e += part.energies
        f += part.forces
        shards = model.__dict__["_spatial_cache"][1]
        assert shards.n_owned == 40500 and shards._tile_hint_owned == _lib.MLP_FLAG_SHAPED
    assert float((f - whole.forces).abs().max()) < 2e-6
    assert abs(float(e - whole.energies)) < 1e-7 * sp.numel()


def test_queued_per_species_launches_on_two_streams(dev):
    """From four rounds of tiles on, the per-species launches of the fused kernel draw their tiles from a queue and alternate
    between the caller's stream and a second one (csrc/mlp.hip).  139 968 water atoms: the eager step (queue + two streams), the same
    step on a side stream of the caller's, and the step captured into a HIP graph (plain launches: a capture stays a chain of kernel
    nodes) give the same energies and forces; per-atom energies are bit-identical (a tile's result does not depend on who computes
    it or when), forces within the float-atomic noise of the AEV backward."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from bench import water_box

    sp_np, x_np, cell_np = water_box(36)   # 139 968 atoms: 2187 tiles of 64
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    model = get_model("ani2x", 0, dev, neighborlist="cell", row_capacity=160)
    assert model._tile_hint(sp, model._elem_idxs(sp).to(torch.int32), sp.numel()) == _lib.MLP_FLAG_SHAPED
    a = model.energies_and_forces(sp, x, cell, pbc)
    b = model.energies_and_forces(sp, x, cell, pbc)
    assert torch.equal(a.atomic_energies, b.atomic_energies) and float((a.forces - b.forces).abs().max()) < 1e-6
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        c = model.energies_and_forces(sp, x, cell, pbc)
    torch.cuda.current_stream(dev).wait_stream(side)
    assert torch.equal(a.atomic_energies, c.atomic_energies) and float((a.forces - c.forces).abs().max()) < 1e-6
    graph = model.graphed(sp, x, cell, pbc)
    g = graph(x)
    assert torch.equal(a.atomic_energies, g.atomic_energies) and float((a.forces - g.forces).abs().max()) < 1e-6
    assert abs(float(a.energies - g.energies)) < 1e-9 * sp.numel()


@pytest.mark.parametrize("pair", [(0, 3), (0, 1), (2, 3), (4, 5), (1, 1)])
def test_skinny_layer0_backward_for_every_slab_count(dev, pair):
    """k_gemm_l0b (layer-0 backward of row tiles with <= 6 flagged AEV slabs) against the row-major hand-over + k_gemm_h2 on
    two-species systems whose species share a radial slab or not (4 / 5 flagged slabs; one species: 2): the same sums in
    the same order, bit for bit.  (With 4 column blocks the kernel's first MFMA of a k step read a register one wait
    state behind the inline-assembly instruction that wrote it -- tools/isa_hazards.py.)"""
    from bench import water_box

    sp_np, x_np, cell_np = water_box(18)   # 17 496 atoms: 256-row tiles from 16 384 on
    x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    base = torch.from_numpy(sp_np).to(dev)
    sp = torch.where(base == 0, pair[0], pair[1]).to(torch.int32)
    model = get_model("ani2x", 0, dev, neighborlist="cell")
    eng = model.aev_computer.engine()
    packed = model.neural_networks._pack(dev)
    nbrs = eng.neighbors(sp, x, cell, (True, True, True), mode="cell", row_cap=160)
    mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
    aev = eng.forward(sp, nbrs, slab_mask=mask)
    got = {}
    try:
        for name, flags in (("skinny", 0), ("rows", _lib.MLP_FLAG_D0_ROWS)):
            packed.flags = flags
            ga = torch.zeros_like(aev)
            e, _, _ = packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask)
            got[name] = (e.clone(), ga.clone())
    finally:
        packed.flags = None
    n_slabs = bin(int(mask[0].item()) & 0xFFFFFFFF).count("1")
    assert n_slabs == (2 if pair[0] == pair[1] else 4 if pair[0] >> 1 == pair[1] >> 1 else 5)
    assert float(got["rows"][1].abs().max()) > 1e-4
    assert torch.equal(got["skinny"][0], got["rows"][0])
    assert torch.equal(got["skinny"][1], got["rows"][1])


@pytest.mark.parametrize("elements", [(0, 3), (0, 1), (1, 1), (0, 2, 6), (0, 1, 2, 3), (0, 1, 2, 3, 4, 5, 6)])
def test_layer0_backward_inside_the_fused_kernel(dev, elements):
    """Phase 5 of k_mlp_fused (round 4: a workgroup owns a tile through all members and adds the members' d E / d AEV in
    place, default from 65 536 atoms) against the d act0 hand-over + layer-0 backward GEMM, forced on a 17 496-atom box for
    1 .. 7 elements (2 .. 32 flagged slabs per tile: one to eight passes of four column blocks): identical per-atom
    energies (the forward phases are the same arithmetic), d E / d AEV within 2e-6 of its largest entry (members summed one
    after the other instead of inside one K = M x H1 reduction), and the whole path against the fp64 oracle."""
    from bench import water_box
    from oracle.sampled_parity import sampled_parity
    from torchani_amd.engine import PackedNetworks

    sp_np, x_np, cell_np = water_box(18)
    rs = np.random.RandomState(7 + sum(elements))
    sp_np = rs.choice(np.asarray(elements), size=sp_np.shape).astype(sp_np.dtype)
    x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    sp64 = torch.from_numpy(sp_np).to(dev)
    sp = sp64.to(torch.int32)
    model = get_model("ani2x", 31, dev, neighborlist="cell", row_capacity=192)
    eng = model.aev_computer.engine()
    packed = model.neural_networks._pack(dev)
    nbrs = eng.neighbors(sp, x, cell, (True, True, True), mode="cell", row_cap=192)
    mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
    aev = eng.forward(sp, nbrs, slab_mask=mask)
    got = {}
    try:
        for name, flags in (("inside", _lib.MLP_FLAG_FUSED_L0B), ("gemm", _lib.MLP_FLAG_NO_FUSED_L0B)):
            packed.flags = flags
            ga = torch.zeros_like(aev)
            e, _, _ = packed.forward_backward(sp, aev, grad_aev=ga, slab_mask=mask)
            got[name] = (e.clone(), ga.clone())
    finally:
        packed.flags = None
    scale = float(got["gemm"][1].abs().max())
    assert scale > 1e-4
    assert torch.equal(got["inside"][0], got["gemm"][0])
    err = float((got["inside"][1] - got["gemm"][1]).abs().max())
    report(f"l0b   elements {elements}: max|d(dE/dAEV)| = {err:.2e} of {scale:.2e}")
    assert err < 2e-6 * scale
    # the whole path (neighbors, AEV, networks with phase 5, AEV backward) against the oracle
    old = PackedNetworks.default_flags
    try:
        PackedNetworks.default_flags = _lib.MLP_FLAG_FUSED_L0B
        out = model.energies_and_forces(sp64, x, cell, (True, True, True), check_overflow=True)
    finally:
        PackedNetworks.default_flags = old
    res = sampled_parity(sp64, x, cell, out.atomic_energies, out.forces, seeded_state("ani2x", 8, 31), "ani2x", 8,
                         n_sample=16, seed=5)
    fmax = max(1.0, float(out.forces.abs().max()))
    report(f"l0b   elements {elements}: max|e_atom err| = {res['max_dE_atom']:.2e}  max|F err| = {res['max_dF']:.2e} (largest force {fmax:.1f})")
    assert res["max_dE_atom"] < 1e-6 * max(1.0, float(out.atomic_energies.abs().max())) and res["max_dF"] < 5e-6 * fmax


def test_two_product_backward_is_off_by_default_and_inside_the_parity_gate(dev):
    """ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS (``model.two_product_backward``; OFF by default): the backward GEMMs of the
    large-system network kernel leave out (weight lo) x (gradient hi).  Energies are bit-identical to the default's, d E / d AEV
    differs by ~2^-12 relative in places -- forces inside north_star's 1e-4 Ha/A against the oracle, and measurably OUTSIDE
    what the default achieves, which is why it is not the default."""
    from bench import water_box
    from oracle.sampled_parity import sampled_parity

    sp_np, x_np, cell_np = water_box(28)   # 65 856 atoms: the layer-0 backward runs inside the fused kernel
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    model = get_model("ani2x", 31, dev, neighborlist="cell", row_capacity=192)
    assert model.two_product_backward is False
    ref = model.energies_and_forces(sp, x, cell, (True, True, True), check_overflow=True)
    model.two_product_backward = True
    try:
        out = model.energies_and_forces(sp, x, cell, (True, True, True), check_overflow=True)
    finally:
        model.two_product_backward = False
    assert torch.equal(out.atomic_energies, ref.atomic_energies)
    diff = float((out.forces - ref.forces).abs().max())
    sd = seeded_state("ani2x", 8, 31)
    res2 = sampled_parity(sp, x, cell, out.atomic_energies, out.forces, sd, "ani2x", 8, n_sample=24, seed=3)
    res3 = sampled_parity(sp, x, cell, ref.atomic_energies, ref.forces, sd, "ani2x", 8, n_sample=24, seed=3)
    report(f"bwd2  two-product backward: max|dF| against the oracle {res2['max_dF']:.2e} Ha/A (three products: {res3['max_dF']:.2e}); "
           f"max|F2 - F3| = {diff:.2e}")
    assert 0.0 < diff and res2["max_dF"] < 1e-4 and res3["max_dF"] < 5e-6 * max(1.0, float(ref.forces.abs().max()))


def test_forward_backward_workspace_is_what_the_call_touches(dev):
    """anihip_mlp_forward_backward_workspace_bytes (ABI 10): a call runs in EXACTLY the bytes the query reports -- behind
    them a guard region keeps its pattern -- for the three shapes of the call: layer-0 backward inside the fused kernel
    (forced here; the default from 65 536 atoms: ~100 B per atom), the d act0 hand-over (8 KB per atom of the first
    hidden layer only), and no gradient at all; one byte less is refused.  The general query stays an upper bound."""
    import ctypes as C

    from bench import water_box

    sp_np, x_np, cell_np = water_box(18)
    sp = torch.from_numpy(sp_np).to(dev).to(torch.int32)
    x, cell = torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    model = get_model("ani2x", 31, dev, neighborlist="cell", row_capacity=192)
    eng = model.aev_computer.engine()
    packed = model.neural_networks._pack(dev)
    nbrs = eng.neighbors(sp, x, cell, (True, True, True), mode="cell", row_cap=192)
    aev = eng.forward(sp, nbrs)
    n = sp.numel()
    L = _lib.lib()
    full = L.anihip_mlp_workspace_bytes(C.byref(packed.desc), n)
    ref_e = None
    try:
        for name, flags, want_grad in (("inside", _lib.MLP_FLAG_FUSED_L0B, True), ("hand-over", _lib.MLP_FLAG_NO_FUSED_L0B, True),
                                       ("energies only", 0, False)):
            packed.desc.flags = flags
            need = L.anihip_mlp_forward_backward_workspace_bytes(C.byref(packed.desc), n, int(want_grad))
            assert 0 < need <= full
            if name == "inside":
                assert need < 400 * n + (1 << 20)      # index lists, tile table, per-member energies
            if name == "hand-over":
                assert 8 * 256 * 4 * n <= need < 8 * 256 * 4 * n + 400 * n + (4 << 20)   # + d E / d act0 of 8 members
            guard = 4096
            ws = torch.full((need + guard,), 0xA5, dtype=torch.uint8, device=dev)
            ae = torch.zeros(n, dtype=torch.float32, device=dev)
            ga = torch.zeros_like(aev) if want_grad else None
            args = lambda nbytes: (torch.cuda.current_stream().cuda_stream, C.byref(packed.desc), n, 0, n, sp.data_ptr(), aev.data_ptr(), None, ws.data_ptr(),
                                   nbytes, ae.data_ptr(), ga.data_ptr() if want_grad else None, None)
            assert L.anihip_mlp_forward_backward(*args(need - 1)) != 0 and b"workspace too small" in L.anihip_last_error()
            _lib.check(L.anihip_mlp_forward_backward(*args(need)))
            torch.cuda.synchronize()
            assert bool((ws[need:] == 0xA5).all()), name
            if ref_e is None:
                ref_e = ae.clone()
            assert torch.equal(ae, ref_e), name       # (the forward phases are the same arithmetic in all three)
            report(f"ws    {name:14s}: {need / n:8.1f} B per atom (general query: {full / n:.1f})")
    finally:
        packed.desc.flags = 0
        packed.flags = None


def test_layer0_backward_inside_the_fused_kernel_is_for_celu_networks(dev):
    """The GELU / bias-free networks of the ANI-2xr family keep the d act0 hand-over at every size (their phase-5
    instantiation spilled registers and was removed in round 5): forcing ANIHIP_MLP_FLAG_FUSED_L0B on them is refused
    loudly, the default path runs."""
    from bench import water_box
    from torchani_amd.models import ANI2xr

    sp_np, x_np, cell_np = water_box(10)
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    model = ANI2xr(seed=3, device=dev, periodic_table_index=False, neighborlist="cell", row_capacity=192)
    model.set_enabled("repulsion_xtb", False)
    ref = model.energies_and_forces(sp, x, cell, (True, True, True), check_overflow=True)
    assert torch.isfinite(ref.forces).all()
    old = PackedNetworks.default_flags
    try:
        PackedNetworks.default_flags = _lib.MLP_FLAG_FUSED_L0B
        with pytest.raises(RuntimeError, match="CELU"):
            model.energies_and_forces(sp, x, cell, (True, True, True))
    finally:
        PackedNetworks.default_flags = old


@pytest.mark.parametrize("periodic", [True, False])
def test_locality_sort_of_a_shuffled_system(dev, periodic):
    """ANI.locality_sort = "auto": a large single system given in an incoherent atom order is evaluated on a cell-sorted
    copy (spatial machinery with one rank) -- same energies and forces, atom by atom, as the coherent order gives; a
    lattice-ordered input is left alone."""
    from bench import water_box

    sp_np, x_np, cell_np = water_box(28)   # 65 856 atoms
    n = sp_np.shape[1]
    perm = np.random.RandomState(5).permutation(n)
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    sp_s, x_s = torch.from_numpy(sp_np[:, perm]).to(dev), torch.from_numpy(np.ascontiguousarray(x_np[:, perm])).to(dev)
    pbc = (True, True, True)
    if not periodic:   # (a droplet: no cell, open boundaries)
        cell, pbc = None, None
    model = get_model("ani2x", 0, dev, neighborlist="cell", row_capacity=192)   # (its own instance)
    try:
        model.locality_sort = "never"
        ref = model.energies_and_forces(sp, x, cell, pbc)
        model.locality_sort = "auto"
        assert not model._wants_locality_sort(sp.to(torch.int32), x, cell, pbc, sp)
        assert model._wants_locality_sort(sp_s.to(torch.int32), x_s, cell, pbc, sp_s)
        out = model.energies_and_forces(sp_s, x_s, cell, pbc)
        assert model.last_collective["n_owned"] == n and model.last_collective["n_halo"] == 0   # (went through the sorted copy)
        p = torch.from_numpy(perm).to(dev)
        assert float((out.forces - ref.forces[:, p]).abs().max()) < 2e-6
        assert float((out.atomic_energies - ref.atomic_energies[:, p]).abs().max()) < 5e-7
        assert abs(float(out.energies - ref.energies)) < 1e-7 * n
        # moved coordinates (a new tensor): the sorted order is kept, results still right
        moved = x_s + 0.02 * torch.randn_like(x_s)
        a = model.energies_and_forces(sp_s, moved, cell, pbc)
        model.locality_sort = "never"
        b = model.energies_and_forces(sp_s, moved, cell, pbc)
        assert float((a.forces - b.forces).abs().max()) < 2e-6 and abs(float(a.energies - b.energies)) < 1e-7 * n
    finally:
        model.locality_sort = "auto"


def test_present_species_first_relabelling(dev):
    """models.ANI.compact_species: a large system's species are numbered "present ones first" inside the engine (water under
    ANI-2x: one radial AEV slab instead of two half-empty ones) with the first-layer weights permuted to match.  Same
    energies and forces as with the reference numbering (the sums run over the slabs in a different order: fp32
    rounding), one flagged slab fewer, and the sharded path adds up to the same."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from bench import water_box

    sp_np, x_np, cell_np = water_box(20)   # 24 000 atoms
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    model = get_model("ani2x", 0, dev, neighborlist="cell", row_capacity=160)   # (its own instance)
    try:
        model.compact_species = False
        ref = model.energies_and_forces(sp, x, cell, pbc)
        model.compact_species = True
        sp32 = sp.to(torch.int32)
        sp_e, order = model._engine_species(sp32)
        assert order == (0, 3, 1, 2, 4, 5, 6)
        assert sorted(torch.unique(sp_e).tolist()) == [0, 1]
        out = model.energies_and_forces(sp, x, cell, pbc)
        assert float((out.atomic_energies - ref.atomic_energies).abs().max()) < 5e-7
        assert float((out.forces - ref.forces).abs().max()) < 2e-6
        assert abs(float(out.energies - ref.energies)) < 1e-7 * sp.numel()
        # one flagged slab fewer per atom
        eng = model.aev_computer.engine()
        for species_e, want in ((sp32, 5.0), (sp_e, 4.0)):
            nbrs = eng.neighbors(species_e, x, cell, pbc, mode="cell", row_cap=160)
            mask = torch.zeros(sp.numel(), dtype=torch.int32, device=dev)
            eng.forward(species_e, nbrs, slab_mask=mask)
            pop = mask.to(torch.int64) & 0xFFFFFFFF
            n_slabs = float(sum(((pop >> b) & 1).double().mean() for b in range(32)))
            assert abs(n_slabs - want) < 0.05, (n_slabs, want)
        # shards of the relabelled system add up to the whole
        e = torch.zeros(1, dtype=torch.float64, device=dev)
        f = torch.zeros_like(x)
        for rank in range(2):
            part = model.energies_and_forces(sp, x, cell, pbc, shard=(rank, 2))
            e += part.energies
            f += part.forces
        assert float((f - ref.forces).abs().max()) < 2e-6 and abs(float(e - ref.energies)) < 1e-7 * sp.numel()
        # three elements spread over three radial slabs (H, N, Cl -> two slabs), with padding atoms in the system
        sp3 = torch.where(sp == 0, 0, 2)
        sp3[0, ::7] = 6
        sp3[0, 5::97] = -1
        model.compact_species = False
        ref3 = model.energies_and_forces(sp3, x, cell, pbc)
        model.compact_species = True
        assert model._engine_species(sp3.to(torch.int32))[1] == (0, 2, 6, 1, 3, 4, 5)
        out3 = model.energies_and_forces(sp3, x, cell, pbc)
        assert float((out3.forces - ref3.forces).abs().max()) < 2e-6
        assert float((out3.atomic_energies - ref3.atomic_energies).abs().max()) < 5e-7
        assert bool((out3.forces[sp3 < 0] == 0).all()) and bool((out3.atomic_energies[sp3 < 0] == 0).all())
        # a HIP graph of the step owns its relabelled species (and the matching weight pack): replays agree, also after
        # another system has taken the model's species cache
        graph = model.graphed(sp, x, cell, pbc)
        assert graph.species_order == (0, 3, 1, 2, 4, 5, 6)
        other = torch.where(sp == 0, 1, 2)
        model.energies_and_forces(other, x, cell, pbc)
        moved = x + 0.01
        replay = graph(moved)
        eager = model.energies_and_forces(sp, moved, cell, pbc)
        assert float((replay.forces - eager.forces).abs().max()) < 2e-6
        assert abs(float(replay.energies - eager.energies)) < 1e-7 * sp.numel()
    finally:
        model.compact_species = True


def test_partition_skin_reuses_the_shards_while_atoms_move(dev):
    """MD use of the spatial shards: with partition_skin the partition is cut once and kept while no atom has moved more
    than skin / 2; the shards' partial results must still add up to the whole-system result at the moved coordinates, and
    a larger move must cut a new partition."""
    box, n = 36.0, 4000
    rs = np.random.RandomState(11)
    x0 = torch.from_numpy(rs.uniform(0, box, (1, n, 3)).astype(np.float32)).to(dev)
    sp = torch.from_numpy(rs.choice([0, 1, 2, 3], size=(1, n), p=[0.5, 0.3, 0.1, 0.1])).to(dev)
    # (a random gas has close contacts; the comparison is shard-sum against whole, both through the same kernels)
    cell = torch.eye(3, device=dev) * box
    pbc = (True, True, True)
    model = get_model("ani1x", 0, dev, neighborlist="cell", row_capacity=320)   # (its own instance: the skin is set on it)
    model.partition_skin = 1.0
    world = 3
    parts = []
    for step, scale in enumerate((0.0, 0.3, 0.45, 2.0)):
        d = torch.from_numpy(rs.normal(size=(1, n, 3)).astype(np.float32)).to(dev)
        x = x0 + scale * d / d.norm(dim=-1, keepdim=True)
        whole = model.energies_and_forces(sp, x, cell, pbc)
        e = torch.zeros(1, dtype=torch.float64, device=dev)
        f = torch.zeros_like(x)
        for rank in range(world):
            out = model.energies_and_forces(sp, x, cell, pbc, shard=(rank, world), check_overflow=True)
            e += out.energies
            f += out.forces
            if rank == world - 1:
                parts.append(model.__dict__["_spatial_cache"][1])
        assert float((f - whole.forces).abs().max()) < 2e-4 * max(1.0, float(whole.forces.abs().max()))
        assert abs(float(e - whole.energies)) < 1e-6 * n
    # (the cache holds one rank's partition at a time, so within a step every rank cuts its own; across steps the last
    # rank's partition is the one the next step's first lookup sees -- rank differs, so it is cut again: check the
    # single-rank behaviour directly)
    # The decision is taken WITHOUT a host synchronisation, one step late (SpatialShards.check_async / poll): a step queues
    # the flags for its coordinates and reads the previous step's.
    model.__dict__.pop("_spatial_cache", None)
    sp32 = sp.view(-1).int()
    a = model._spatial_partition(sp32, x0.view(-1, 3), cell, pbc, 1, world)
    x1 = x0 + 0.3
    x1[..., 1:] -= 0.3
    b = model._spatial_partition(sp32, x1.view(-1, 3), cell, pbc, 1, world)            # moved 0.3 A: kept
    x2 = x0 + 0.26                                                                       # moved 0.45 A >= 0.8 x skin / 2
    c = model._spatial_partition(sp32, x2.view(-1, 3), cell, pbc, 1, world)            # ... found out one step late: kept
    x3 = x2 + 0.0
    d = model._spatial_partition(sp32, x3.view(-1, 3), cell, pbc, 1, world)            # renewed now
    assert b is a and c is a and d is not a
    # the FIRST moved step behind a cut has no flags of a previous step to go by: it reads its own at once (same-step guard)
    x4 = x3 + 0.4                                                                        # 0.69 A > skin / 2 in ONE step
    e = model._spatial_partition(sp32, x4.view(-1, 3), cell, pbc, 1, world)
    assert e is not d                                                                    # ... renewed in the same step
    # in a steady loop the flags are read one step late: a jump inside ONE step runs on the old halo and the NEXT call raises
    x5 = x4 + 0.01
    f5 = model._spatial_partition(sp32, x5.view(-1, 3), cell, pbc, 1, world)           # (verified at once, flags queued)
    x6 = x5 + 0.4
    f6 = model._spatial_partition(sp32, x6.view(-1, 3), cell, pbc, 1, world)           # the flags of x5 say "fine": kept
    assert f5 is e and f6 is e
    with pytest.raises(RuntimeError, match="partition_skin"):                           # ... and the next call says so
        model._spatial_partition(sp32, (x6 + 0.0).view(-1, 3), cell, pbc, 1, world)
    # ... or check_partition() behind the last step of a loop
    model.__dict__.pop("_spatial_cache", None)
    a = model._spatial_partition(sp32, x0.view(-1, 3), cell, pbc, 1, world)
    model._spatial_partition(sp32, (x0 + 0.01).view(-1, 3), cell, pbc, 1, world)
    assert model._spatial_partition(sp32, (x0 + 0.4).view(-1, 3), cell, pbc, 1, world) is a
    with pytest.raises(RuntimeError, match="partition_skin"):
        model.check_partition()
    # partition_check = "strict": every step reads its own flags before it is evaluated -- never a stale halo
    model.partition_check = "strict"
    a = model._spatial_partition(sp32, x0.view(-1, 3), cell, pbc, 1, world)
    b = model._spatial_partition(sp32, (x0 + 0.01).view(-1, 3), cell, pbc, 1, world)
    c = model._spatial_partition(sp32, (x0 + 0.4).view(-1, 3), cell, pbc, 1, world)
    assert b is a and c is not a
    model.check_partition()
    model.partition_check = "lagged"
    # another species tensor (other padding atoms) never reuses the partition
    model.__dict__.pop("_spatial_cache", None)
    a = model._spatial_partition(sp32, x0.view(-1, 3), cell, pbc, 1, world)
    assert model._spatial_partition(sp32.clone(), x0.view(-1, 3), cell, pbc, 1, world) is not a


@pytest.mark.parametrize("name", ["ch4_ani1x", "rand_batch_ani2x", "water_pbc_ani2x"])
def test_autograd_path_equals_fused(dev, name):
    """model((species, coords)) + torch.autograd == fused engine path (same kernels underneath)."""
    from torchani_amd.grad import energies_and_forces

    g = load_golden(name)
    sp, x, cell, pbc = to_dev(g, dev)
    model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"])
    pbc_t = None if pbc is None else torch.tensor(pbc)
    e, f = energies_and_forces(model, sp, x, cell, pbc_t)
    out = model.energies_and_forces(sp, x, cell, pbc)
    assert not x.requires_grad
    assert np.abs(f.cpu().numpy() - g["forces"]).max() < F_TOL
    assert torch.allclose(f, out.forces, atol=2e-6)
    assert e.requires_grad   # (the reference's tuple: energies keep their graph, grad.py:263-290)
    assert np.abs(e.detach().double().cpu().numpy() - g["energies"]).max() < 2e-6 * np.abs(g["energies"]).max()
    # NN-only energies with the shifter disabled (arch.py:136-142)
    model.set_enabled("energy_shifter", False)
    e_nn = model((sp, x), cell, pbc_t).energies
    model.set_enabled("energy_shifter", True)
    assert np.abs(e_nn.double().cpu().numpy() - g["energies_nn"]).max() < 1e-5 * max(1.0, np.sqrt(sp.shape[1]))


@pytest.mark.parametrize("name", ["rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x"])
def test_hip_graph_replay(dev, name):
    """energies_and_forces captured into a HIP graph: replays reproduce the eager result, also after the
    coordinates change."""
    g = load_golden(name)
    sp, x, cell, pbc = to_dev(g, dev)
    model = get_model(g["kind"], g["seed"], dev, cutoff_fn=g["cutoff_fn"], neighborlist=modes_for(g)[-1], row_capacity=256)
    f = model.graphed(sp, x, cell, pbc)
    out = f(x, cell)
    torch.cuda.synchronize()
    f.check()
    assert np.abs(out.atomic_energies.cpu().numpy() - g["atomic_energies"]).max() < E_ATOM_TOL
    assert np.abs(out.forces.cpu().numpy() - g["forces"]).max() < F_TOL
    # other coordinates through the same graph == eager evaluation of those coordinates
    x2 = x + 0.02 * torch.randn_like(x) * (sp >= 0).unsqueeze(-1)
    ref = model.energies_and_forces(sp, x2, cell, pbc)
    e_ref, f_ref = ref.energies.clone(), ref.forces.clone()
    out2 = f(x2, cell)
    torch.cuda.synchronize()
    assert torch.allclose(out2.energies, e_ref, atol=1e-9, rtol=1e-12)
    assert torch.allclose(out2.forces, f_ref, atol=2e-6)
    # and back
    out3 = f(x, cell)
    assert np.abs(out3.forces.cpu().numpy() - g["forces"]).max() < F_TOL


def test_hip_graph_owns_its_buffers(dev):
    """A captured graph keeps replaying correctly after (a) a LARGER eager call on the same model (which used to free
    and reallocate the networks' workspace under the graph) and (b) an in-place parameter update (the graph re-captures
    with the new weight planes instead of silently using the old ones)."""
    from torchani_amd.models import ANI2x

    g = load_golden("rand_batch_ani2x")
    sp, x, cell, pbc = to_dev(g, dev)
    model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False)
    f = model.graphed(sp, x)
    ref = f(x).forces.clone()
    packed = model.neural_networks._pack(dev)
    ws_ptr = packed._ws.data_ptr()
    assert packed.pinned == 1
    # (a) a much larger eager evaluation: 40x the atoms
    big_sp, big_x = sp.repeat(40, 1), x.repeat(40, 1, 1)
    model.energies_and_forces(big_sp, big_x)
    assert packed._ws.data_ptr() == ws_ptr          # the captured workspace was neither freed nor replaced
    assert torch.equal(f(x).forces, ref) or torch.allclose(f(x).forces, ref, atol=2e-6)
    # (b) in-place update of one layer
    with torch.no_grad():
        lin = next(model.neural_networks.parameters())
        lin.mul_(1.01)
    eager = model.energies_and_forces(sp, x)
    out = f(x)
    assert f.n_captures == 2
    assert torch.allclose(out.energies, eager.energies, atol=1e-9) and torch.allclose(out.forces, eager.forces, atol=2e-6)
    assert not torch.allclose(out.forces, ref, atol=1e-7)   # the update is visible


def test_auto_graph_follows_the_species_tensor_not_its_address(dev):
    """energies_and_forces replays a HIP graph from the third call with the SAME species tensor.  A stream of batches,
    each in a fresh species tensor of the same shape (the caching allocator hands the freed address to the next one), must
    never be served from a graph captured for an earlier batch; an MD-like loop over one tensor still gets its graph."""
    from torchani_amd.models import ANI2x

    g = load_golden("rand_batch_ani2x")
    sp0, x, _, _ = to_dev(g, dev)
    model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False)
    rs = np.random.RandomState(0)
    real = (sp0 >= 0)
    for it in range(6):
        perm = torch.from_numpy(rs.permutation(7)).to(dev)
        sp = torch.where(real, perm[sp0.clamp(min=0)], sp0)      # another batch: same shape, other elements
        got = model.energies_and_forces(sp, x)
        e, f = got.energies.clone(), got.forces.clone()
        model.auto_graph_atoms, keep = 0, model.auto_graph_atoms
        ref = model.energies_and_forces(sp, x)
        model.auto_graph_atoms = keep
        assert torch.allclose(e, ref.energies, atol=1e-9) and torch.allclose(f, ref.forces, atol=2e-6), it
        del sp, got
    assert all(ent[1] is None for ent in model._graphs.values())   # (no tensor was seen three times)
    for _ in range(4):
        out = model.energies_and_forces(sp0, x)
    assert any(ent[1] is not None for ent in model._graphs.values())
    assert np.abs(out.forces.cpu().numpy() - g["forces"]).max() < F_TOL


def test_factories_are_loud_about_random_weights(dev):
    """No state_dict and no seed -> a warning; a state dict that does not match the architecture -> an error
    (never a silent fall-back to random parameters)."""
    from torchani_amd.models import ANI2x

    with pytest.warns(UserWarning, match="RANDOM parameters"):
        ANI2x(device=dev, n_members=1)
    with pytest.raises(RuntimeError, match="does not provide"):
        ANI2x(device=dev, state_dict={"unrelated.weight": np.zeros(3, dtype=np.float32)})


def test_ensemble_values_are_differentiable(dev):
    """Autograd through ensemble_values=True (nn/_containers.py:638-651 is differentiable in the reference): the gradient
    of a weighted sum of the member energies equals the weighted sum of the single-member models' gradients."""
    g = load_golden("rand_batch_ani2x")
    sp, x, _, _ = to_dev(g, dev)
    model = get_model(g["kind"], g["seed"], dev)
    model.set_enabled("energy_shifter", False)
    w = torch.linspace(-1.0, 2.0, 8, device=dev)
    xs = x.clone().requires_grad_(True)
    em = model((sp, xs), ensemble_values=True).energies          # [M, C]
    assert em.shape == (8, sp.shape[0])
    (gx,) = torch.autograd.grad((em * w[:, None]).sum(), xs)
    ref = torch.zeros_like(x)
    for m in range(8):
        xm = x.clone().requires_grad_(True)
        e = model[m]((sp, xm)).energies
        (gm,) = torch.autograd.grad(e.sum(), xm)
        ref += w[m] * gm
        assert torch.allclose(e.detach(), em[m].detach(), atol=2e-6)
    model.set_enabled("energy_shifter", True)
    assert (gx - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item())


def test_api_details(dev):
    """Legacy tuple call, atomic / ensemble_values outputs, active members (nn/_containers.py:590-660)."""
    g = load_golden("simple2_ani2x")
    sp, x, _, _ = to_dev(g, dev)
    model = get_model("ani2x", g["seed"], dev)
    with pytest.warns(UserWarning):
        s2, aev = model.aev_computer((sp, x))
    assert aev.shape == (2, 7, 1008) and s2 is sp
    em = model.neural_networks(sp, aev, atomic=True, ensemble_values=True)
    ea = model.neural_networks(sp, aev, atomic=True)
    assert em.shape == (8, 2, 7) and torch.allclose(em.mean(0), ea, atol=1e-6)  # tests/test_ensemble.py:20-36
    assert np.abs(em.cpu().numpy() - g["member_atomic_energies"]).max() < E_ATOM_TOL
    ens = model.neural_networks
    ens.set_active_members([1, 5])
    assert ens.get_active_members_num() == 2
    e2 = ens(sp, aev, atomic=True)
    ens.set_active_members(list(range(8)))
    assert torch.allclose(e2, (em[1] + em[5]) / 2, atol=1e-6)
    with pytest.raises(ValueError):
        model.aev_computer(sp.cpu(), x.cpu())
    # atomic numbers in, element indices out; unsupported element -> ValueError
    from torchani_amd.models import ANI2x

    m2 = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev)
    znum = torch.tensor([1, 6, 7, 8, 16, 9, 17], device=dev)
    z = torch.where(sp >= 0, znum[sp.clamp(min=0)], torch.full_like(sp, -1))
    out_z = m2.energies_and_forces(z, x)
    out_i = model.energies_and_forces(sp, x)
    assert torch.equal(out_z.energies, out_i.energies)
    with pytest.raises(ValueError):
        m2.energies_and_forces(torch.full_like(sp, 5), x)


def test_ensemble_conveniences(dev):
    """atomic_energies / energies_qbcs / atomic_stdev / members_forces / force_magnitudes / force_qbc /
    model[i] / len(model) of the reference model (arch.py:133-135,245-264,385-585), against the
    reference-generated member energies of the golden case and the golden ensemble forces."""
    g = load_golden("simple2_ani2x")
    sp, x, _, _ = to_dev(g, dev)
    model = get_model("ani2x", g["seed"], dev)
    me = g["member_atomic_energies"]                               # [8, C, A] network part
    sae = np.asarray(model.energy_shifter.self_energies.cpu(), dtype=np.float64)
    real = g["species"] >= 0
    sae_mol = np.where(real, sae[np.clip(g["species"], 0, None)], 0.0).sum(1)
    e_members = me.sum(-1) + sae_mol                               # [8, C]
    assert len(model) == 8
    # atomic energies (with self energies, like the reference's forward(atomic=True))
    _, ea = model.atomic_energies((sp, x))
    ref_atomic = np.where(real, me.mean(0) + sae[np.clip(g["species"], 0, None)], 0.0)
    assert np.abs(ea.double().cpu().numpy() - ref_atomic).max() < 2e-5 * np.abs(ref_atomic).max()  # fp32 output
    # qbc factors
    _, e, qbc = model.energies_qbcs((sp, x))
    n_at = real.sum(1)
    assert np.abs(qbc.double().cpu().numpy() - e_members.std(0, ddof=1) / np.sqrt(n_at)).max() < 2e-4
    assert np.abs(e.double().cpu().numpy() - e_members.mean(0)).max() < 2e-5 * np.abs(e_members).max()
    # atomic stdev
    _, _, sd = model.atomic_stdev((sp, x))
    assert np.abs(sd.double().cpu().numpy() - np.where(real, me.std(0, ddof=1), 0.0)).max() < 1e-5
    # members' forces: their mean is the ensemble force; members' energies match the reference
    _, em, fm = model.members_forces((sp, x))
    assert fm.shape == (8,) + tuple(x.shape) and em.shape == (8, sp.shape[0])
    assert np.abs(fm.mean(0).cpu().numpy() - g["forces"]).max() < F_TOL
    assert np.abs(em.cpu().numpy() - e_members).max() < 1e-5
    assert list(model.neural_networks.active_members_idxs) == list(range(8))   # restored
    # single-member view
    m3 = model[3]
    o3 = m3.energies_and_forces(sp, x)
    assert torch.allclose(o3.forces, fm[3], atol=2e-6) and torch.allclose(o3.energies, em[3], atol=1e-9)
    # force statistics
    _, mags = model.force_magnitudes((sp, x))
    assert torch.allclose(mags, fm.norm(dim=-1).mean(0), atol=1e-6)
    _, mg, rstd, rrange = model.force_qbc((sp, x))
    allm = fm.norm(dim=-1)
    assert torch.allclose(rstd, (allm.std(0) + 1e-8) / (allm.mean(0) + 1e-8), atol=1e-5)
    assert torch.all(rrange >= rstd)


def _water_box(n_side, seed, dev, box_per=3.1):
    """n_side^3 waters on a jittered lattice, periodic cubic box (0.1 atoms / A^3)."""
    rs = np.random.RandomState(seed)
    L = n_side * box_per
    gx = (np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) * box_per
    o = gx + rs.uniform(-0.3, 0.3, gx.shape)
    def rnd_unit(n):
        v = rs.normal(size=(n, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    a = rnd_unit(len(o))
    b = np.cross(a, rnd_unit(len(o)))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    th = np.deg2rad(104.52) / 2
    h1 = o + 0.9572 * (np.cos(th) * a + np.sin(th) * b)
    h2 = o + 0.9572 * (np.cos(th) * a - np.sin(th) * b)
    x = np.stack([o, h1, h2], 1).reshape(1, -1, 3).astype(np.float32)
    sp = np.tile(np.array([3, 0, 0]), len(o)).reshape(1, -1)
    cell = np.eye(3, dtype=np.float32) * L
    return sp, x, cell


def test_water_box_properties_and_oracle(dev, oracle64):
    """Mid-size periodic box (3000 atoms): cell-list rows == all-pairs rows, oracle parity, Newton's 3rd
    law, translation / image invariance (size-independent properties used again at bench sizes)."""
    sp, x, cell = _water_box(10, 1, dev)
    model_c = get_model("ani2x", 21, dev, neighborlist="cell")
    model_b = get_model("ani2x", 21, dev, neighborlist="batch")
    spt, xt, ct = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
    pbc = (True, True, True)
    oc = model_c.energies_and_forces(spt, xt, ct, pbc, check_overflow=True)
    ob = model_b.energies_and_forces(spt, xt, ct, pbc, check_overflow=True)
    assert torch.allclose(oc.atomic_energies, ob.atomic_energies, atol=2e-6)
    assert torch.allclose(oc.forces, ob.forces, atol=5e-6)
    dims, flat, sae = oracle_networks("ani2x", 8, 21)
    ref = oracle64.energy_forces(oracle_params("ani2x"), sp, x.astype(np.float64), dims, flat, 8, sae=sae,
                                 cell=cell, pbc=pbc, cell_list=True)
    ea = np.abs(oc.atomic_energies.cpu().numpy() - ref["atomic_energies"]).max()
    fe = np.abs(oc.forces.cpu().numpy() - ref["forces"]).max()
    et = abs(float(oc.energies[0]) - ref["energies"][0])
    report(f"box   water 3000 atoms pbc        max|e_atom err| = {ea:.2e}  |E_total err| = {et:.2e}  |F err| = {fe:.2e}")
    assert ea < E_ATOM_TOL and fe < F_TOL and et < 1e-4
    assert oc.forces.sum(dim=1).abs().max() < 2e-4  # momentum conservation
    # translate by a non-lattice vector and by a lattice vector: energies unchanged
    sh = torch.tensor([1.234, -7.77, 30.05], device=dev)
    o2 = model_c.energies_and_forces(spt, xt + sh, ct, pbc)
    o3 = model_c.energies_and_forces(spt, xt + ct[1], ct, pbc)
    assert abs(float(o2.energies[0] - oc.energies[0])) < 2e-4
    assert abs(float(o3.energies[0] - oc.energies[0])) < 2e-4
    assert torch.allclose(o3.forces, oc.forces, atol=2e-5)


def test_solvated_box_46k(dev, oracle64):
    """BASELINE config 3 scale: ~46k-atom periodic water box.  Full-size checks are the size-independent
    properties; a random sample of atoms is checked against the oracle on the same box."""
    sp, x, cell = _water_box(25, 3, dev)  # 15625 waters = 46875 atoms, 77.5 A box
    model = get_model("ani2x", 23, dev, neighborlist="cell")
    spt, xt, ct = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
    pbc = (True, True, True)
    out = model.energies_and_forces(spt, xt, ct, pbc, check_overflow=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out.forces).all()
    assert out.forces.sum(dim=1).abs().max() < 2e-3
    # oracle on the same box: AEVs of every atom, networks on a random sample of 1500 atoms
    dims, flat, _ = oracle_networks("ani2x", 8, 23)
    aev = oracle64.aev(oracle_params("ani2x"), sp, x.astype(np.float64), cell, pbc, cell_list=True)
    pick = np.random.RandomState(0).choice(sp.shape[1], 1500, replace=False)
    ae, _, _ = oracle64.mlp(sp[0, pick], aev[0, pick], dims, flat, n_members=8, want_grad=False)
    ea = np.abs(out.atomic_energies.cpu().numpy()[0, pick] - ae).max()
    report(f"box   water 46875 atoms pbc       max|e_atom err| (1500 sampled atoms) = {ea:.2e}")
    assert ea < E_ATOM_TOL
    # forces: d E / d aev of every atom from the oracle's networks, back through the oracle's AEV backward
    _, ga, _ = oracle64.mlp(sp, aev, dims, flat, n_members=8)
    _, gc = oracle64.aev(oracle_params("ani2x"), sp, x.astype(np.float64), cell, pbc, cell_list=True, grad_aev=ga)
    fe = np.abs(out.forces.cpu().numpy() + gc).max()
    report(f"box   water 46875 atoms pbc       max|F err| (all atoms) = {fe:.2e}")
    assert fe < F_TOL


@pytest.mark.parametrize("name", ["cfg3_1hz5_water_ani2x", "cfg3_1c17_ani2x"])
def test_config3_reference_inputs(dev, name):
    """BASELINE config 3 on the reference's own structures (tests/golden/gen_golden_configs.py): 1hz5 solvated to 46 357
    atoms in a periodic box (7 species present in one system: H C N O S + water), and 1C17.pdb (16 649 atoms, no PBC).
    Per-atom energies AND forces on a 2000-atom sample, totals, against the reference's fp64 values."""
    from _util import load_sampled

    g = load_sampled(name)
    model = get_model("ani2x", g["seed"], dev, neighborlist="cell")
    sp = torch.from_numpy(g["species"]).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    cell = None if g["cell"] is None else torch.from_numpy(g["cell"]).to(dev)
    pbc = None if g["pbc"] is None else tuple(bool(b) for b in g["pbc"])
    out = model.energies_and_forces(sp, x, cell, pbc)
    torch.cuda.synchronize()
    pick = g["sample"]
    ea = np.abs(out.atomic_energies.cpu().numpy()[0, pick] - g["atomic_energies_sample"]).max()
    fe = np.abs(out.forces.cpu().numpy()[0, pick] - g["forces_sample"]).max()
    n = sp.shape[1]
    de = abs(float(out.energies[0]) - float(g["energies"][0]))
    report(f"cfg3  {name:22s} {n} atoms  max|e_atom err| = {ea:.2e}  |F err| = {fe:.2e}  |E err| = {de:.2e}")
    assert ea < E_ATOM_TOL and fe < F_TOL
    assert de < E_ATOM_TOL * np.sqrt(n)
    f64 = out.forces.double()
    assert abs(float((f64 ** 2).sum()) - float(g["force_sq_sum"])) < 1e-5 * float(g["force_sq_sum"])
    # the autograd path on the same input agrees (other kernels' launch configuration: whole rows, no slab masks)
    xs = x.clone().requires_grad_(True)
    pbc_t = None if pbc is None else torch.tensor(pbc)
    e = model((sp, xs), cell, pbc_t).energies
    (gx,) = torch.autograd.grad(e.sum(), xs)
    assert np.abs(-gx.cpu().numpy()[0, pick] - g["forces_sample"]).max() < F_TOL


@pytest.mark.parametrize("case", ["rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x", "triclinic_pbc_ani2x",
                                  "cos_water_pbc_ani2x"])
def test_xtb_repulsion_matches_reference(dev, case):
    """RepulsionXTB (potentials/xtb.py) on the engine's neighbor rows against the reference's fp64 values
    (tests/golden/gen_golden_pairs.py): per-atom halves, molecular energies, forces; alone, through the model's autograd
    path and inside energies_and_forces (NN + pair term), incl. a pair cutoff beyond the AEV's radial cutoff."""
    from torchani_amd.models import ANI2x
    from torchani_amd.potentials import RepulsionXTB

    name = case[4:] if case.startswith("cos_") else case
    g = load_golden(name)
    ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"pairs_{case}.npz")))
    sp, x, cell, pbc = to_dev(g, dev)
    model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False,
                  neighborlist=modes_for(g)[-1], row_capacity=256)
    pot = RepulsionXTB(g["symbols"], cutoff=float(ref["cutoff"]), cutoff_fn=str(ref["cutoff_fn"])).to(dev)
    sp32 = sp.to(torch.int32).contiguous()
    rows = model._pair_rows(pot, sp32, x, cell, pbc)   # rows with the pair potential's own cutoff
    n = sp32.numel()
    ae = torch.zeros(n, dtype=torch.float32, device=dev)
    gc = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    pot.accumulate(sp32, rows, ae, gc)
    torch.cuda.synchronize()
    rows.raise_on_overflow()
    escale = max(1.0, np.abs(ref["atomic_energies"]).max())
    fscale = max(1.0, np.abs(ref["forces"]).max())
    ea = np.abs(ae.cpu().numpy().reshape(ref["atomic_energies"].shape) - ref["atomic_energies"]).max()
    fe = np.abs(-gc.cpu().numpy().reshape(ref["forces"].shape) - ref["forces"]).max()
    report(f"xtb   {case:24s} max|e_atom err| = {ea:.2e} (scale {escale:.1f})  |F err| = {fe:.2e} (scale {fscale:.1f})")
    assert ea < 2e-6 * escale and fe < 5e-6 * fscale
    # inside the model: NN + repulsion
    model.add_pair_potential("repulsion_xtb", pot)
    out = model.energies_and_forces(sp, x, cell, pbc)
    e_ref = g["energies"] + ref["energies"]
    f_ref = g["forces"] + ref["forces"]
    assert np.abs(out.energies.cpu().numpy() - e_ref).max() < 1e-5 * max(1.0, np.abs(e_ref).max() * 1e-2) + 2e-6 * escale * sp.shape[1]
    assert np.abs(out.forces.cpu().numpy() - f_ref).max() < F_TOL + 5e-6 * fscale
    xs = x.clone().requires_grad_(True)
    pbc_t = None if pbc is None else torch.tensor(pbc)
    e = model((sp, xs), cell, pbc_t).energies
    (gx,) = torch.autograd.grad(e.sum(), xs)
    assert np.abs(-gx.cpu().numpy() - f_ref).max() < F_TOL + 5e-6 * fscale
    # per-atom energies with the pair halves (core.py:195-198), under no_grad
    with torch.no_grad():
        ea_model = model((sp, x), cell, pbc_t, atomic=True).energies
    ea_ref = torch.from_numpy(g["energies"])   # molecular reference: compare the sums, and the pair part per atom
    assert np.abs(ea_model.double().sum(dim=1).cpu().numpy() - e_ref).max() < 1e-5 * max(1.0, np.abs(e_ref).max() * 1e-2) + 2e-6 * escale * sp.shape[1] + 2e-7 * np.abs(e_ref).max()
    with pytest.raises(NotImplementedError):
        model((sp, x.clone().requires_grad_(True)), cell, pbc_t, atomic=True)
    # switched off again (arch.py:136-142 set_enabled)
    model.set_enabled("repulsion_xtb", False)
    out0 = model.energies_and_forces(sp, x, cell, pbc)
    assert np.abs(out0.forces.cpu().numpy() - g["forces"]).max() < F_TOL


@pytest.mark.parametrize("case", ["rand_batch_ani2x", "water_pbc_ani2x", "triclinic_pbc_ani2x"])
def test_analytic_pair_potentials_match_reference(dev, case):
    """RepulsionZBL, LennardJones / RepulsionLJ / DispersionLJ, FixedCoulomb and FixedMNOK (potentials/zbl.py, lj.py,
    fixed_coulomb.py) on the engine's neighbor rows against the reference's own classes in fp64
    (tests/golden/gen_golden_pairs2.py, same constructor arguments): per-atom halves and forces, relative to the scale of
    each potential (the random geometries make Lennard-Jones forces of 1e4 Ha/A)."""
    import importlib.util

    from torchani_amd import potentials as P
    from torchani_amd.models import ANI2x

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("gen_pairs2_consts", os.path.join(gdir, "gen_golden_pairs2.py"))
    src = open(spec.origin).read().split("def cases(symbols):")[0].split("CHARGES = ")[1]
    ns: dict = {}
    exec("CHARGES = " + src, ns)   # the element constants of the generator (its imports need the reference)
    g = load_golden(case)
    ref = dict(np.load(os.path.join(gdir, f"pairs2_{case}.npz")))
    sp, x, cell, pbc = to_dev(g, dev)
    symbols = list(g["symbols"])
    q = tuple(ns["CHARGES"][s] for s in symbols)
    pots = {
        "zbl": P.RepulsionZBL(symbols, cutoff=5.2, cutoff_fn="smooth"),
        "zbl_cos": P.RepulsionZBL(symbols, k=0.4685, cutoff=4.0, cutoff_fn="cosine"),
        "lj": P.LennardJones(symbols, eps=tuple(ns["EPS"][s] for s in symbols), sigma=tuple(ns["SIGMA"][s] for s in symbols),
                             cutoff=7.5, cutoff_fn="smooth"),
        "lj_rep": P.RepulsionLJ(symbols, cutoff=5.2, cutoff_fn="smooth"),
        "lj_disp": P.DispersionLJ(symbols, cutoff=7.5, cutoff_fn="smooth"),
        "coulomb": P.FixedCoulomb(symbols, charges=q, dielectric=1.3, cutoff=7.5, cutoff_fn="smooth"),
        "mnok": P.FixedMNOK(symbols, charges=q, eta=tuple(ns["ETA"][s] for s in symbols), cutoff=7.5, cutoff_fn="smooth"),
    }
    model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False,
                  neighborlist=modes_for(g)[-1], row_capacity=256)
    sp32 = sp.to(torch.int32).contiguous()
    n = sp32.numel()
    for key, pot in pots.items():
        pot = pot.to(dev)
        rows = model._pair_rows(pot, sp32, x, cell, pbc)
        ae = torch.zeros(n, dtype=torch.float32, device=dev)
        gc = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        pot.accumulate(sp32, rows, ae, gc)
        torch.cuda.synchronize()
        rows.raise_on_overflow()
        ea_ref, f_ref = ref[key + "_atomic"], ref[key + "_forces"]
        escale, fscale = max(1e-3, np.abs(ea_ref).max()), max(1e-3, np.abs(f_ref).max())
        ea = np.abs(ae.cpu().numpy().reshape(ea_ref.shape) - ea_ref).max()
        fe = np.abs(-gc.cpu().numpy().reshape(f_ref.shape) - f_ref).max()
        report(f"pair2 {case:20s} {key:8s} max|e_atom err| = {ea:.2e} (scale {escale:.1e})  |F err| = {fe:.2e} (scale {fscale:.1e})")
        # (fp32: the 12th power alone carries ~12 roundings of the distance)
        assert ea < 5e-6 * escale and fe < 1e-5 * fscale, key
    # through the model's autograd path: networks + one of them
    model.add_pair_potential("zbl", pots["zbl"].to(dev))
    xs = x.clone().requires_grad_(True)
    pbc_t = None if pbc is None else torch.tensor(pbc)
    e = model((sp, xs), cell, pbc_t).energies
    (gx,) = torch.autograd.grad(e.sum(), xs)
    f_ref = g["forces"] + ref["zbl_forces"]
    assert np.abs(-gx.cpu().numpy() - f_ref).max() < F_TOL + 1e-5 * np.abs(ref["zbl_forces"]).max()
    with pytest.raises(ValueError):
        P.FixedCoulomb(symbols, charges=q[:-1])


@pytest.mark.parametrize("case", ["rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x", "triclinic_pbc_ani2x",
                                  "b973c_water_pbc_ani2x"])
def test_d3_dispersion_matches_reference(dev, case):
    """TwoBodyDispersionD3 (potentials/dftd3.py) against the reference's fp64 values (tests/golden/gen_golden_d3.py:
    cutoff 8 A, smooth envelope): per-atom halves, molecular energies, forces including the dependence of C6 on the
    coordination numbers; alone, as a shard (lo .. hi), inside energies_and_forces together with the networks and the
    xTB repulsion (the ANI-2xr recipe, arch.py:1176-1181), through autograd, and the virial by finite strain."""
    from torchani_amd.models import ANI2x
    from torchani_amd.potentials import RepulsionXTB, TwoBodyDispersionD3

    name = case[6:] if case.startswith("b973c_") else case
    g = load_golden(name)
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = dict(np.load(os.path.join(gdir, f"d3_{case}.npz")))
    sp, x, cell, pbc = to_dev(g, dev)
    model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False,
                  neighborlist=modes_for(g)[-1], row_capacity=256)
    pot = TwoBodyDispersionD3.from_functional(g["symbols"], str(ref["functional"]), cutoff=float(ref["cutoff"]),
                                              cutoff_fn=str(ref["cutoff_fn"])).to(dev)
    sp32 = sp.to(torch.int32).contiguous()
    rows = model._pair_rows(pot, sp32, x, cell, pbc)
    n = sp32.numel()
    ae = torch.zeros(n, dtype=torch.float32, device=dev)
    gc = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    pot.accumulate(sp32, rows, ae, gc)
    torch.cuda.synchronize()
    rows.raise_on_overflow()
    escale = max(1e-3, np.abs(ref["atomic_energies"]).max())
    fscale = max(1e-3, np.abs(ref["forces"]).max())
    ea = np.abs(ae.cpu().numpy().reshape(ref["atomic_energies"].shape) - ref["atomic_energies"]).max()
    fe = np.abs(-gc.cpu().numpy().reshape(ref["forces"].shape) - ref["forces"]).max()
    report(f"d3    {case:24s} max|e_atom err| = {ea:.2e} (scale {escale:.1e})  |F err| = {fe:.2e} (scale {fscale:.1e})")
    # 1e-5 of the scale -- or, for the random dense geometries of rand_batch (coordination numbers up to 7.3, far from
    # every reference: the Gaussian reference weights amplify the fp32 rounding of the CNs), 5e-7 Ha / 5e-5 Ha/A, still
    # 20x / 2x inside the parity gates
    assert ea < max(1e-5 * escale, 5e-7) and fe < max(2e-5 * fscale, 5e-5)
    # a shard: energies / gradient rows of lo .. hi only, equal to the same rows of the full evaluation
    lo, hi = n // 3, (2 * n) // 3
    ae2 = torch.zeros_like(ae)
    gc2 = torch.zeros_like(gc)
    pot.accumulate(sp32, rows, ae2, gc2, lo=lo, hi=hi)
    assert torch.equal(ae2[lo:hi], ae[lo:hi]) and torch.equal(gc2[lo:hi], gc[lo:hi])
    assert ae2[:lo].abs().max() == 0 and gc2[hi:].abs().max() == 0
    # inside the model: networks + repulsion + dispersion
    rep = dict(np.load(os.path.join(gdir, f"pairs_{name}.npz")))
    model.add_pair_potential("repulsion_xtb", RepulsionXTB(g["symbols"], cutoff=float(rep["cutoff"]),
                                                           cutoff_fn=str(rep["cutoff_fn"])).to(dev))
    model.add_pair_potential("dispersion_d3", pot)
    out = model.energies_and_forces(sp, x, cell, pbc)
    e_ref = g["energies"] + rep["energies"] + ref["energies"]
    f_ref = g["forces"] + rep["forces"] + ref["forces"]
    assert np.abs(out.energies.cpu().numpy() - e_ref).max() < 1e-5 * max(1.0, np.abs(e_ref).max() * 1e-2) + 1e-5
    assert np.abs(out.forces.cpu().numpy() - f_ref).max() < F_TOL
    xs = x.clone().requires_grad_(True)
    pbc_t = None if pbc is None else torch.tensor(pbc)
    e = model((sp, xs), cell, pbc_t).energies
    (gx,) = torch.autograd.grad(e.sum(), xs)
    assert np.abs(-gx.cpu().numpy() - f_ref).max() < F_TOL
    if cell is not None and all(pbc):
        # virial of the dispersion term alone = strain derivative of its energy (central differences)
        w = torch.zeros((3, 3), dtype=torch.float64, device=dev)
        pot.accumulate(sp32, rows, None, gc2, w)

        def e_of(strain):
            F = torch.eye(3, device=dev, dtype=torch.float64) + strain
            xx = (x.double() @ F.T).float()
            cc = (cell.double() @ F.T).float()
            r2 = model._pair_rows(pot, sp32, xx, cc, pbc)
            a = torch.zeros(n, dtype=torch.float32, device=dev)
            pot.accumulate(sp32, r2, a, None)
            return a.double().sum().item()

        h = 2e-3
        for (a_, b_) in ((0, 0), (1, 2)):
            st = torch.zeros((3, 3), dtype=torch.float64, device=dev)
            st[a_, b_] = h
            fd = (e_of(st) - e_of(-st)) / (2 * h)
            assert abs(fd - w[a_, b_].item()) < 2e-3 * max(abs(fd), 1e-4) + 2e-6, (a_, b_, fd, w[a_, b_].item())


@pytest.mark.parametrize("kind,case", [(k, c) for k in ("ani2xr", "ani2dr")
                                       for c in ("rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x")] +
                         [("anir2s", "rand_batch_ani2x"), ("anir2s", "dense90_ani2x")])
def test_ani2xr_family_matches_reference(dev, kind, case):
    """The ANI-2xr / ANI-2dr architecture (models.py:252-325: simple_ani AEV with the smooth envelope, GELU networks
    without biases, xTB repulsion, D3 dispersion and B97-3c self energies for -2dr) against the reference's own builder in
    fp64 with the same seeded parameters (tests/golden/gen_golden_2xr.py): energies and forces through
    energies_and_forces and through autograd."""
    from torchani_amd.models import ANI2dr, ANI2xr, ANIr2s
    from torchani_amd.weights import random_state_dict

    ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"x2r_{kind}_{case}.npz")))
    sp = torch.from_numpy(ref["species"]).to(dev)
    x = torch.from_numpy(ref["coords"]).to(dev)
    cell = torch.from_numpy(ref["cell"]).to(dev) if "cell" in ref else None
    pbc = tuple(bool(b) for b in ref["pbc"]) if "pbc" in ref else None
    factory = {"ani2xr": ANI2xr, "ani2dr": ANI2dr, "anir2s": ANIr2s}[kind]   # (ANI-r2s: models.py:325-368)
    model = factory(state_dict=random_state_dict(kind, 8, int(ref["seed"])), device=dev, periodic_table_index=False,
                    neighborlist="batch" if cell is None or sp.shape[0] > 1 else "auto", row_capacity=256)
    assert [str(s) for s in ref["symbols"]] == list(model.symbols)
    out = model.energies_and_forces(sp, x, cell, pbc)
    torch.cuda.synchronize()
    n_real = int((ref["species"] >= 0).sum(axis=1).max())
    fscale = max(1.0, np.abs(ref["forces"]).max())
    ee = np.abs(out.energies.cpu().numpy() - ref["energies"]).max()
    fe = np.abs(out.forces.cpu().numpy() - ref["forces"]).max()
    report(f"x2r   {kind} {case:20s} max|E err| = {ee:.2e} ({n_real} atoms)  |F err| = {fe:.2e} (|F|max {fscale:.1f})")
    assert ee < E_ATOM_TOL * n_real and fe < F_TOL * fscale
    xs = x.clone().requires_grad_(True)
    e = model((sp, xs), cell, None if pbc is None else torch.tensor(pbc)).energies
    (gx,) = torch.autograd.grad(e.sum(), xs)
    # (the module path returns energies in the dtype of the coordinates, like the reference: fp32 totals of ~ -2400 Ha)
    assert np.abs(e.detach().cpu().numpy() - ref["energies"]).max() < E_ATOM_TOL * n_real + 2e-7 * np.abs(ref["energies"]).max()
    assert np.abs(-gx.cpu().numpy() - ref["forces"]).max() < F_TOL * fscale
    # a trainable GELU model goes through the training passes (bias-free layers, GELU derivatives from the kept
    # pre-activations): same energies, and the graph reaches the parameters (test_gpu_training.py pins the gradients)
    model.neural_networks.requires_grad_(True)
    et = model((sp, x), cell, None if pbc is None else torch.tensor(pbc)).energies
    assert et.requires_grad
    assert np.abs(et.detach().cpu().numpy() - ref["energies"]).max() < E_ATOM_TOL * n_real + 2e-7 * np.abs(ref["energies"]).max()
    et.sum().backward()
    w = model.neural_networks.members[0].atomics["H"].layers[0].weight
    assert w.grad is not None and torch.isfinite(w.grad).all() and float(w.grad.abs().max()) > 0


@pytest.mark.parametrize("case", ["rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x"])
def test_animbis_charges_match_reference(dev, case):
    """ANI-mbis (models.py:201-252): ANI-2x energies plus atomic charges from the two-output GELU charge networks and the
    electronegativity / hardness normalizer, against the reference's Assembler(cls=ANIq) in fp64 with the same seeded
    parameters (tests/golden/gen_golden_mbis.py).  The charges add up to the total charge exactly like the reference's."""
    from torchani_amd.models import ANImbis
    from torchani_amd.tuples import SpeciesEnergiesAtomicCharges
    from torchani_amd.weights import random_charge_state_dict, random_state_dict

    ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"mbis_{case}.npz")))
    seed = int(ref["seed"])
    sd = dict(random_state_dict("ani2x", 8, seed))
    sd.update({"potentials.nnp.charge_networks." + k: v for k, v in random_charge_state_dict(seed).items()})
    sp = torch.from_numpy(ref["species"]).to(dev)
    x = torch.from_numpy(ref["coords"]).to(dev)
    cell = torch.from_numpy(ref["cell"]).to(dev) if "cell" in ref else None
    pbc = torch.tensor(ref["pbc"]) if "pbc" in ref else None
    model = ANImbis(state_dict=sd, device=dev, periodic_table_index=False,
                    neighborlist="batch" if cell is None or sp.shape[0] > 1 else "auto", row_capacity=256)
    out = model((sp, x), cell, pbc)
    torch.cuda.synchronize()
    assert isinstance(out, SpeciesEnergiesAtomicCharges) and out.atomic_charges.shape == sp.shape
    n_real = int((ref["species"] >= 0).sum(axis=1).max())
    ee = np.abs(out.energies.cpu().numpy() - ref["energies"]).max()
    raw = model.potentials["nnp"].charge_networks(sp, model.aev_computer(sp, x, cell, pbc), atomic=True)
    re = np.abs(raw.cpu().numpy() - ref["raw_charges"]).max()
    qe = np.abs(out.atomic_charges.cpu().numpy() - ref["atomic_charges"]).max()
    report(f"mbis  {case:20s} max|E err| = {ee:.2e} ({n_real} atoms)  |q_raw err| = {re:.2e}  |q err| = {qe:.2e} "
           f"(|q|max {np.abs(ref['atomic_charges']).max():.2f})")
    assert ee < E_ATOM_TOL * n_real + 2e-7 * np.abs(ref["energies"]).max()
    assert re < 2e-5 and qe < 2e-5
    assert out.atomic_charges.sum(dim=1).abs().max().item() < 1e-5
    assert (out.atomic_charges[sp < 0] == 0).all()
    # one member of the ensemble keeps the charge networks (arch.py ANIq.__getitem__ through the Assembler's members)
    one = model[2]((sp, x), cell, pbc)
    assert torch.equal(one.atomic_charges, out.atomic_charges)
    assert torch.equal(model.atomic_charges((sp, x), cell, pbc), out.atomic_charges)


@pytest.mark.parametrize("case", ["chno", "chno1x"])
def test_simple_ani_builder_matches_reference(dev, case):
    """models.simple_ani (arch.py:992-1066) against the reference's builder in fp64 with the same seeded parameters
    (tests/golden/gen_golden_simple.py): a four-element model with the default recipe (ANI-2x widths, GELU, no biases,
    smooth envelope, xTB repulsion, two members) and one with the ANI-1x recipe (CELU with biases, cosine cutoff, 4 x 8
    angular grid, no repulsion)."""
    from torchani_amd.models import simple_ani

    ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"simple_{case}.npz")))
    kw = dict(ensemble_size=2) if case == "chno" else dict(
        ensemble_size=1, container_ctor="like_1x", activation="celu", bias=True, cutoff_fn="cosine", angular_shifts=4,
        sections=8, angular_precision=8.0, angular_zeta=32.0, radial_precision=16.0, repulsion=False)
    model = simple_ani([str(s) for s in ref["symbols"]], "wb97x-631gd", seed=int(ref["seed"]), device=dev,
                       periodic_table_index=False, neighborlist="batch", **kw)
    assert model.aev_computer.engine().L == int(ref["aev_dim"])
    assert np.allclose(model.energy_shifter.self_energies.cpu().numpy(), ref["self_energies"])
    sp = torch.from_numpy(ref["species"]).to(dev)
    x = torch.from_numpy(ref["coords"]).to(dev)
    out = model.energies_and_forces(sp, x)
    torch.cuda.synchronize()
    n_real = int((ref["species"] >= 0).sum(axis=1).max())
    fscale = max(1.0, np.abs(ref["forces"]).max())
    ee = np.abs(out.energies.cpu().numpy() - ref["energies"]).max()
    fe = np.abs(out.forces.cpu().numpy() - ref["forces"]).max()
    report(f"simple_ani {case:7s} max|E err| = {ee:.2e} ({n_real} atoms)  |F err| = {fe:.2e} (|F|max {fscale:.2f})")
    assert ee < E_ATOM_TOL * n_real and fe < F_TOL * fscale
    if case == "chno1x":   # CELU with biases: the autograd / training path works on a builder-made model too
        xs = x.clone().requires_grad_(True)
        e = model((sp, xs)).energies
        (gx,) = torch.autograd.grad(e.sum(), xs)
        assert np.abs(-gx.cpu().numpy() - ref["forces"]).max() < F_TOL * fscale


def test_simple_aniq_builder_matches_reference(dev):
    """models.simple_aniq (arch.py:1069-1185: separate charge networks + electronegativity / hardness normalizer) against the
    reference's builder in fp64 with the same seeded parameters; dipoles from the charges (electro.py) in the three
    reference frames (tests/golden/gen_golden_simple.py)."""
    from torchani_amd.constants import HIDDEN_DIMS_2X
    from torchani_amd.extras.electro import compute_dipole
    from torchani_amd.models import simple_aniq
    from torchani_amd.weights import NN_PREFIX, random_network_state_dict

    ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simple_chnoq.npz")))
    sym = [str(s) for s in ref["symbols"]]
    seed, hidden = int(ref["seed"]), {s: HIDDEN_DIMS_2X[s] for s in sym}
    sd = dict(random_network_state_dict(sym, 384, hidden, 2, seed, False))
    sd.update({"potentials.nnp.charge_networks." + k[len(NN_PREFIX):]: v
               for k, v in random_network_state_dict(sym, 384, hidden, 1, 1000 + seed, False, scale=3.0).items()})
    model = simple_aniq(sym, "wb97x-631gd", ensemble_size=2, state_dict=sd, device=dev, periodic_table_index=False,
                        neighborlist="batch")
    sp = torch.from_numpy(ref["species"]).to(dev)
    x = torch.from_numpy(ref["coords"]).to(dev)
    out = model((sp, x))
    torch.cuda.synchronize()
    n_real = int((ref["species"] >= 0).sum(axis=1).max())
    ee = np.abs(out.energies.cpu().numpy() - ref["energies"]).max()
    qe = np.abs(out.atomic_charges.cpu().numpy() - ref["atomic_charges"]).max()
    report(f"simple_aniq chnoq   max|E err| = {ee:.2e} ({n_real} atoms)  |q err| = {qe:.2e} "
           f"(|q|max {np.abs(ref['atomic_charges']).max():.3f})")
    assert ee < E_ATOM_TOL * n_real + 2e-7 * np.abs(ref["energies"]).max() and qe < 5e-6
    znum = torch.from_numpy(ref["atomic_numbers"]).to(dev)
    for frame in ("center_of_mass", "center_of_geometry", "origin"):
        mu = compute_dipole(znum, x.double(), torch.from_numpy(ref["atomic_charges"]).to(dev), frame)
        assert np.abs(mu.cpu().numpy() - ref["dipole_" + frame]).max() < 1e-9, frame
    with pytest.raises(ValueError):
        simple_aniq(sym, "wb97x-631gd", merge_charge_networks=True)


def test_pair_potential_called_on_its_own(dev):
    """``potential(species, coords)`` without a model (core.py:37-67): the potential builds its own neighbor rows.  Same
    reference values as the model-embedded evaluation (pairs2_*.npz, ZBL): per-atom halves with ``atomic=True``, molecular
    energies in float64, forces through autograd."""
    from torchani_amd import potentials as P

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    case = "rand_batch_ani2x"
    g = load_golden(case)
    sp, x, cell, pbc = to_dev(g, dev)
    ref2 = dict(np.load(os.path.join(gdir, f"pairs2_{case}.npz")))
    pot = P.RepulsionZBL(list(g["symbols"]), cutoff=5.2, cutoff_fn="smooth").to(dev)
    ea = pot(sp, x, cell, pbc, atomic=True, atomic_nums_input=False)
    xs = x.clone().requires_grad_(True)
    e = pot(sp, xs, cell, pbc, atomic_nums_input=False)
    (gx,) = torch.autograd.grad(e.sum(), xs)
    torch.cuda.synchronize()
    ea_ref, f_ref = ref2["zbl_atomic"], ref2["zbl_forces"]
    escale, fscale = max(1e-3, np.abs(ea_ref).max()), max(1e-3, np.abs(f_ref).max())
    assert np.abs(ea.cpu().numpy().reshape(ea_ref.shape) - ea_ref).max() < 5e-6 * escale
    assert np.abs(-gx.cpu().numpy().reshape(f_ref.shape) - f_ref).max() < 2e-5 * fscale
    assert e.dtype == torch.float64
    assert np.abs(e.detach().cpu().numpy() - ea_ref.reshape(sp.shape).sum(axis=1)).max() < 2e-5 * escale * sp.shape[1]


def test_periodic_replica_and_symmetries_at_scale(dev):
    """Size-independent properties at ~0.33 M atoms (no oracle at this size): a periodic box replicated 2 x 2 x 2
    has the same per-atom energies and forces as the original box (every atom sees the same environment), the
    net force vanishes, and a rigid translation / a permutation of the atoms change nothing."""
    sp, x, cell = _water_box(24, 11, dev)            # 41 472 atoms
    n = sp.shape[1]
    model = get_model("ani2x", 5, dev, neighborlist="cell")
    pbc = (True, True, True)
    spt, xt, ct = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
    base = model.energies_and_forces(spt, xt, ct, pbc, check_overflow=True)
    e0, f0, a0 = base.energies.clone(), base.forces.clone(), base.atomic_energies.clone()
    # 2 x 2 x 2 replica: 331 776 atoms
    L = np.diag(cell).astype(np.float32)
    shifts = np.array([[i, j, k] for i in range(2) for j in range(2) for k in range(2)], dtype=np.float32) * L
    x8 = (x[0][None, :, :] + shifts[:, None, :]).reshape(1, 8 * n, 3)
    sp8 = np.tile(sp[0], 8).reshape(1, 8 * n)
    big = model.energies_and_forces(torch.from_numpy(sp8).to(dev), torch.from_numpy(x8).to(dev),
                                    torch.from_numpy(2 * cell).to(dev), pbc, check_overflow=True)
    torch.cuda.synchronize()
    a8 = big.atomic_energies.view(8, n)
    f8 = big.forces.view(8, n, 3)
    ea = (a8 - a0.view(1, n)).abs().max().item()
    fe = (f8 - f0.view(1, n, 3)).abs().max().item()
    report(f"box   water replica 8 x {n} atoms   max|e_atom - e_atom(box)| = {ea:.2e}  |F - F(box)| = {fe:.2e}")
    assert ea < 2e-6 and fe < 2e-5      # (coordinates differ by fp32 rounding of the shifted positions)
    assert abs(big.energies.item() - 8 * e0.item()) < 1e-6 * abs(8 * e0.item())
    assert big.forces.sum(dim=1).abs().max() < 2e-2 and f0.sum(dim=1).abs().max() < 2e-3
    # rigid translation (wraps atoms across the cell boundary) and a permutation of the atom order
    t = torch.tensor([1.2345, -7.25, 40.0], device=dev)
    tr = model.energies_and_forces(spt, xt + t, ct, pbc)
    assert (tr.forces - f0).abs().max() < 2e-5 and abs(tr.energies.item() - e0.item()) < 1e-6 * abs(e0.item())
    perm = torch.from_numpy(np.random.RandomState(1).permutation(n)).to(dev)
    pm = model.energies_and_forces(spt[:, perm], xt[:, perm], ct, pbc)
    assert (pm.forces - f0[:, perm]).abs().max() < 2e-5
    assert (pm.atomic_energies - a0[:, perm]).abs().max() < 2e-6


def test_degenerate_inputs(dev, oracle64):
    """Edge cases the reference tests as well (tests/test_aev.py padding / isolated atoms): an isolated atom, atoms
    beyond the cutoff, an all-padding molecule, species that do not occur, and the row-capacity overflow report."""
    model = get_model("ani2x", 11, dev)
    dims, flat, sae = oracle_networks("ani2x", 8, 11)
    p = oracle_params("ani2x")
    sp = np.array([[0, -1, -1, -1],      # one isolated H
                   [3, 0, -1, -1],       # O and H 9 A apart: no neighbors at all
                   [-1, -1, -1, -1],     # nothing
                   [1, 0, 0, 6]], dtype=np.int64)   # a small cluster incl. Cl
    x = np.zeros((4, 4, 3), dtype=np.float32)
    x[1, 1] = [9.0, 0.0, 0.0]
    x[3] = [[0, 0, 0], [1.0, 0.1, 0], [-0.4, 0.9, 0.2], [0.3, -0.5, 1.6]]
    ref = oracle64.energy_forces(p, sp, x.astype(np.float64), dims, flat, 8, sae=sae)
    out = model.energies_and_forces(torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev), check_overflow=True)
    torch.cuda.synchronize()
    assert np.abs(out.energies.cpu().numpy() - ref["energies"]).max() < 1e-6
    assert np.abs(out.atomic_energies.cpu().numpy() - ref["atomic_energies"]).max() < E_ATOM_TOL
    assert np.abs(out.forces.cpu().numpy() - ref["forces"]).max() < F_TOL
    f = out.forces.cpu().numpy()
    assert np.all(f[0] == 0) and np.all(f[1] == 0) and np.all(f[2] == 0)
    assert out.energies[2].item() == 0.0           # padding contributes nothing, not even self energies
    # the autograd path agrees on the same input
    xs = torch.from_numpy(x).to(dev).requires_grad_(True)
    e = model((torch.from_numpy(sp).to(dev), xs)).energies
    (gx,) = torch.autograd.grad(e.sum(), xs)
    assert np.abs(-gx.cpu().numpy() - ref["forces"]).max() < F_TOL
    # more neighbors than the row capacity: reported, never silently truncated (csrc/aev.cu:229 asserts instead)
    rs = np.random.RandomState(0)
    dense = rs.uniform(0, 3.2, (1, 60, 3)).astype(np.float32)
    spd = torch.zeros((1, 60), dtype=torch.int64, device=dev)
    from torchani_amd.models import ANI2x

    small = ANI2x(state_dict=seeded_state("ani2x", 8, 11), device=dev, periodic_table_index=False, row_capacity=16)
    big = get_model("ani2x", 11, dev, row_capacity=256)
    with pytest.warns(UserWarning, match="retrying with 256"):   # checked by default: one retry at the largest capacity
        o16 = small.energies_and_forces(spd, torch.from_numpy(dense).to(dev))
    o256 = big.energies_and_forces(spd, torch.from_numpy(dense).to(dev))
    assert small.aev_computer.row_capacity == 256 and torch.equal(o16.energies, o256.energies)
    # ... and an error when even the largest rows cannot hold an atom's neighbors (here > 128 inside the angular cutoff)
    blob = rs.uniform(0, 1.9, (1, 300, 3)).astype(np.float32)
    spb = torch.zeros((1, 300), dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError, match="overflow"):
        big.energies_and_forces(spb, torch.from_numpy(blob).to(dev))
    with pytest.raises(RuntimeError, match="overflow"):   # the autograd path checks too
        big((spb, torch.from_numpy(blob).to(dev)))
    # unchecked on request (graph capture, latency-critical loops): the status word is still there
    big.energies_and_forces(spb, torch.from_numpy(blob).to(dev), check_overflow=False)
    assert big.aev_computer.last_neighbors().overflowed()


def test_external_neighbors_molecule_idxs(dev):
    """_molecule_idxs of compute_from_external_neighbors (arch.py:171-206): two molecules that overlap in space, given
    as ONE conformation with an all-pairs list, interact only inside each molecule."""
    g = load_golden("rand_batch_ani2x")
    model = get_model("ani2x", g["seed"], dev)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    real0, real1 = (sp[0] >= 0), (sp[1] >= 0)
    sp_cat = torch.cat([sp[0][real0], sp[1][real1]]).unsqueeze(0)
    x_cat = torch.cat([x[0][real0], x[1][real1]]).unsqueeze(0)
    n0, n = int(real0.sum()), sp_cat.shape[1]
    mol = torch.cat([torch.zeros(n0, dtype=torch.long), torch.ones(n - n0, dtype=torch.long)]).to(dev)
    i, j = torch.triu_indices(n, n, offset=1, device=dev)
    pairs = torch.stack([i, j])
    e_joint = model.compute_from_external_neighbors(sp_cat, x_cat, pairs, None, _molecule_idxs=mol).energies
    e_sep = model.energies_and_forces(sp[:2], x[:2]).energies
    # (the autograd path returns float32 totals incl. self energies like the reference: one ulp at 3.5 kHa = 2.4e-4)
    assert abs(e_joint.item() - e_sep.sum().item()) < 1e-3
    e_all = model.compute_from_external_neighbors(sp_cat, x_cat, pairs, None).energies
    assert abs(e_all.item() - e_joint.item()) > 1e-2      # the molecules do overlap: the filter matters
    with pytest.raises(ValueError, match="same length"):
        model.compute_from_external_neighbors(sp_cat, x_cat, pairs, None, _molecule_idxs=mol[:-1])


def test_grad_helpers(dev):
    """torchani_amd.grad mirrors torchani/grad.py: single_point dictionary (incl. ensemble statistics), leaf checks of
    forces(), forces_for_training (differentiable once more), hessians unsupported."""
    from torchani_amd import grad

    g = load_golden("simple2_ani2x")
    sp, x, _, _ = to_dev(g, dev)
    model = get_model("ani2x", g["seed"], dev)
    out = grad.single_point(model, sp, x, forces=True, atomic_energies=True)
    assert set(out) == {"energies", "atomic_energies", "forces"} and not x.requires_grad
    assert np.abs(out["forces"].cpu().numpy() - g["forces"]).max() < F_TOL
    assert np.abs(out["energies"].double().cpu().numpy() - g["energies"]).max() < 2e-6 * np.abs(g["energies"]).max()
    ens = grad.single_point(model, sp, x, ensemble_values=True)
    assert ens["ensemble_values"].shape == (8, sp.shape[0]) and ens["qbcs"].shape == (sp.shape[0],)
    assert torch.allclose(ens["energies"], out["energies"], atol=1e-3)
    n_at = (sp >= 0).sum(dim=1).float()
    assert torch.allclose(ens["qbcs"], ens["ensemble_values"].std(0, unbiased=True) / n_at.sqrt())
    with pytest.raises(ValueError, match="require grad"):
        grad.forces(out["energies"], x)
    with pytest.raises(NotImplementedError):
        grad.single_point(model, sp, x, hessians=True)
    # forces_for_training: a graph that reaches the parameters
    from torchani_amd.models import ANI2x

    m2 = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False)
    m2.neural_networks.requires_grad_(True)
    xs = x.clone().requires_grad_(True)
    e = m2((sp, xs)).energies
    f = grad.forces_for_training(e, xs)
    assert f.requires_grad
    f.pow(2).sum().backward()
    w = m2.neural_networks.members[0].atomics["H"].layers[0].weight
    assert w.grad is not None and torch.isfinite(w.grad).all() and w.grad.abs().max() > 0


def test_energies_are_run_to_run_deterministic(dev):
    """neighbor rows, AEVs and per-atom energies are bit-reproducible from run to run (no atomics on
    that path); forces accumulate with float atomics and may differ in the last bits."""
    from bench import water_box

    sp, x, cell = water_box(12)
    model = get_model("ani2x", 13, dev, neighborlist="cell")
    spt = torch.from_numpy(sp.astype(np.int64)).to(dev)
    xt, ct = torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
    runs = [model.energies_and_forces(spt, xt, ct, (True, True, True)) for _ in range(3)]
    torch.cuda.synchronize()
    for r in runs[1:]:
        assert torch.equal(r.atomic_energies, runs[0].atomic_energies)
        assert torch.equal(r.energies, runs[0].energies)
        assert (r.forces - runs[0].forces).abs().max().item() < 1e-6
    aevc = model.aev_computer
    a = [aevc(spt, xt, ct, torch.tensor([True, True, True])) for _ in range(2)]
    assert torch.equal(a[0], a[1])
    # deterministic_forces: int64 fixed-point accumulation (ANIHIP_BWD_FIXED_POINT) -> forces and virial-free results
    # are bit-identical from run to run, and agree with the float-atomic path to accumulation round-off
    from torchani_amd.models import ANI2x

    det = ANI2x(state_dict=seeded_state("ani2x", 8, 13), device=dev, periodic_table_index=False, neighborlist="cell")
    det.deterministic_forces = True
    druns = [det.energies_and_forces(spt, xt, ct, (True, True, True)) for _ in range(4)]
    torch.cuda.synchronize()
    for r in druns[1:]:
        assert torch.equal(r.forces, druns[0].forces) and torch.equal(r.energies, druns[0].energies)
    assert (druns[0].forces - runs[0].forces).abs().max().item() < 1e-6
    # (a different decomposition is a different -- equally valid -- rounding: a pair that straddles two shards is
    # pushed as two separately rounded terms instead of one gathered sum; reproducibility is per decomposition)
    parts = [det._energies_and_forces_core(spt.to(torch.int32), xt, ct, (True, True, True), None, True, False, (r, 3))
             for r in range(3)]
    assert (sum(p.forces for p in parts) - druns[0].forces).abs().max().item() < 1e-7


def test_member_model_keeps_pair_potentials(dev):
    """ANI2dr(...)[k] (and model_index=k) evaluates repulsion and dispersion like the full model restricted to member k
    (the reference deep-copies the whole model, arch.py:252-261)."""
    from torchani_amd.models import ANI2dr

    g = load_golden("small_ani2x") if "small_ani2x" in GOLDEN_NAMES else load_golden(GOLDEN_NAMES[0])
    sp, x, cell, pbc = to_dev(g, dev)
    full = ANI2dr(seed=0, device=dev, periodic_table_index=False)
    member = full[3]
    assert set(member.potentials.keys()) == {"nnp", "repulsion_xtb", "dispersion_d3"}
    a = member.energies_and_forces(sp, x, cell, pbc)
    bare = member.energies_and_forces(sp, x, cell, pbc)   # (second call: same path, warmed)
    full.set_active_members([3])
    b = full.energies_and_forces(sp, x, cell, pbc)
    full.set_active_members(list(range(8)))
    assert torch.allclose(a.energies, b.energies, atol=1e-9) and torch.allclose(a.forces, b.forces, atol=1e-7)
    assert torch.equal(a.energies, bare.energies)
    member.set_enabled("repulsion_xtb", False)
    member.set_enabled("dispersion_d3", False)
    c = member.energies_and_forces(sp, x, cell, pbc)
    member.set_enabled("repulsion_xtb", True)
    member.set_enabled("dispersion_d3", True)
    assert (a.energies - c.energies).abs().max() > 1e-6, "the pair potentials did not contribute"


def test_aev_with_unequally_spaced_shifts(dev, oracle64):
    """AEVComputer constants whose shifts are NOT equally spaced (from_constants accepts any, aev/_computer.py:602-666):
    anihip_aev_table_pack leaves ANIHIP_AEV_UNIFORM_SHFA (1: the forward's three-exponential recurrence over ShfA) and / or
    ANIHIP_AEV_REC_BWD (2: the backward's recurrences over ShfR and ShfA) clear and the kernels evaluate every Gaussian
    directly; equally spaced shifts set both.  AEV and its VJP against the oracle for every combination."""
    from oracle import oracle as orc
    from torchani_amd.aev import AEVComputer
    from torchani_amd.weights import arch_spec

    g = load_golden("rand_batch_ani2x")
    base = arch_spec("ani2x")[1]
    sp, x, cell, pbc = to_dev(g, dev)
    C, A = g["species"].shape
    w_np = np.random.RandomState(77).uniform(-1.0, 1.0, (C, A, base.out_dim)).astype(np.float32)
    w = torch.from_numpy(w_np).to(dev)
    bent_a = tuple(0.8 + 0.3375 * k + 0.05 * (k % 3) for k in range(8))
    bent_r = tuple(v + (0.02 if k % 5 == 2 else 0.0) for k, v in enumerate(base.ShfR))
    for shfr, shfa, want_flags in ((base.ShfR, bent_a, 0), (base.ShfR, base.ShfA, 3), (bent_r, base.ShfA, 1)):
        consts = base._replace(ShfA=tuple(float(v) for v in shfa), ShfR=tuple(float(v) for v in shfr))
        aevc = AEVComputer(consts, neighborlist="batch", row_capacity=256).to(dev)
        xx = x.clone().requires_grad_(True)
        aev_t = aevc(sp, xx, cell, None)
        (vjp,) = torch.autograd.grad((aev_t * w).sum(), xx)
        aev = aev_t.detach().cpu().numpy()
        assert (aevc.engine().params.flags & 3) == want_flags
        p = orc.make_params(7, consts.Rcr, consts.Rca, consts.EtaR, consts.EtaA, consts.Zeta, consts.ShfR, consts.ShfA,
                            consts.ShfZ, "cosine")
        ref, ref_vjp = oracle64.aev(p, g["species"], g["coords"].astype(np.float64), grad_aev=w_np.astype(np.float64))
        err = np.abs(aev - ref).max()
        verr, vmag = np.abs(vjp.cpu().numpy() - ref_vjp).max(), np.abs(ref_vjp).max()
        report(f"aev   custom shifts (flags={want_flags})   max|aev err| = {err:.2e}   vjp err = {verr:.2e} (|vjp|max {vmag:.1f})")
        assert err < AEV_TOL
        assert verr <= VJP_REG_REL * max(1.0, vmag)


def _stress_state(case, seed=11):
    """ANI-2x x 8 parameter sets that stress the split-fp16 network arithmetic (its power-of-two scales come from
    weight-norm BOUNDS, include/anihip.h: anihip_mlp_desc.fused_bounds) away from the uniform +-1/sqrt(fan_in) init."""
    from torchani_amd.weights import NN_PREFIX, random_state_dict

    sd = {k: v.copy() for k, v in random_state_dict("ani2x", 8, seed).items()}
    rs = np.random.RandomState(seed + 1)
    layer_of = lambda k: 3 if ".final_layer." in k else int(k.split(".layers.")[1].split(".")[0])   # noqa: E731
    if case.startswith("scale"):
        sc = {"scale_small": (0.125, 0.125, 0.125, 0.125), "scale_large": (8.0, 8.0, 8.0, 8.0),
              "scale_mixed": (8.0, 0.125, 8.0, 0.125)}[case]
        for k in sd:
            if k.startswith(NN_PREFIX):
                sd[k] = (sd[k] * np.float32(sc[layer_of(k)])).astype(np.float32)
    elif case == "student_t":
        for k in sd:
            if k.startswith(NN_PREFIX) and k.endswith("weight"):
                bound = 1.0 / np.sqrt(sd[k].shape[1])
                sd[k] = (rs.standard_t(3, size=sd[k].shape) * bound / np.sqrt(3.0)).astype(np.float32)
    elif case == "outlier_row":
        for sym, layer, r in (("H", 1, 7), ("O", 0, 100), ("C", 2, 3)):
            k = f"{NN_PREFIX}members.3.atomics.{sym}.layers.{layer}.weight"
            sd[k][r] *= np.float32(100.0)
    else:
        raise ValueError(case)
    return sd


@pytest.mark.parametrize("case", ["scale_small", "scale_large", "scale_mixed", "student_t", "outlier_row"])
@pytest.mark.parametrize("name", ["rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x"])
def test_network_arithmetic_under_weight_distribution_stress(dev, oracle64, name, case):
    """The f16x3 (split-fp16 MFMA) ensemble against the fp64 oracle for parameters that are NOT the benchmark's uniform
    init: per-layer scales 1/8 and 8 (and alternating), a heavy-tailed Student-t(3) init, one member with rows 100 x
    larger.  Gates: the north_star's 1e-5 Ha / 1e-4 Ha/A, relative to the largest reference value when that exceeds 1
    (an 8 x per layer scale multiplies the energies by ~4000: no fp32 path holds 1e-5 Ha absolute there, the fp32
    reference included)."""
    from oracle import oracle as orc
    from torchani_amd.models import ANI2x
    from torchani_amd.weights import arch_spec

    g = load_golden(name)
    sd = _stress_state(case)
    sp, x, cell, pbc = to_dev(g, dev)
    model = ANI2x(state_dict=sd, device=dev, periodic_table_index=False)
    out = model.energies_and_forces(sp, x, cell, pbc)
    symbols, _, _ = arch_spec("ani2x")
    dims, flat = orc.pack_networks(sd, symbols, 8)
    ref = oracle64.energy_forces(oracle_params("ani2x"), g["species"], g["coords"].astype(np.float64), dims, flat, 8,
                                 sae=None, cell=g["cell"], pbc=pbc)
    e_ref, f_ref = ref["atomic_energies"], ref["forces"]
    real = g["species"] >= 0
    ea = np.abs(out.atomic_energies.cpu().numpy() - e_ref)[real].max()
    fe = np.abs(out.forces.cpu().numpy() - f_ref)[real].max()
    emag, fmag = np.abs(e_ref[real]).max(), np.abs(f_ref[real]).max()
    report(f"stress {case:12s} {name:18s} |e_atom err| = {ea:.2e} (|e|max {emag:.2e})  |F err| = {fe:.2e} (|F|max {fmag:.2e})")
    assert ea <= E_ATOM_TOL * max(1.0, emag) and fe <= F_TOL * max(1.0, fmag)


def test_headline_scale_sampled_parity(dev):
    """>= 1 M atoms (several network chunks, 32-bit row offsets, bin-local coordinates in a 220 A box): 256 sampled atoms of
    a 1 073 733-atom periodic water box against the fp64 oracle on the 10.2 A clusters around them (oracle/sampled_parity.py;
    bench.py runs the same check at 2.34 M atoms after its timed loop)."""
    from bench import water_box
    from oracle.sampled_parity import sampled_parity

    sp_np, x_np, cell_np = water_box(71)   # 71^3 waters
    model = get_model("ani2x", 23, dev, neighborlist="cell")
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    out = model.energies_and_forces(sp, x, cell, (True, True, True), check_overflow=True)
    assert sp.numel() > 1_000_000 and torch.isfinite(out.forces).all()
    res = sampled_parity(sp, x, cell, out.atomic_energies, out.forces, seeded_state("ani2x", 8, 23), "ani2x", 8,
                         n_sample=256, seed=3)
    report(f"box   water {sp.numel()} atoms pbc  sampled n={res['n']}: max|e_atom err| = {res['max_dE_atom']:.2e}  "
           f"max|F err| = {res['max_dF']:.2e}  ({res['seconds']:.0f} s oracle)")
    assert res["max_dE_atom"] < E_ATOM_TOL and res["max_dF"] < F_TOL
    assert res["max_dE_atom"] <= E_ATOM_REG and res["max_dF"] <= F_REG, "regression gate (module header)"
    # size-independent properties at this size: no net force on a periodic system (every pair and triple term pushes its
    # atoms with forces that cancel), and the same answer for the box translated by a vector that is no multiple of anything
    # (other bins, other bin-local coordinates, other images)
    n = sp.numel()
    net = out.forces.double().sum(dim=1).abs().max()
    fsum = out.forces.double().abs().sum()
    shift = torch.tensor([1.234, -2.5, 3.75], dtype=x.dtype, device=dev)
    moved = model.energies_and_forces(sp, x + shift, cell, (True, True, True), check_overflow=True)
    dE = abs(float(moved.energies - out.energies))
    dF = float((moved.forces - out.forces).abs().max())
    report(f"box   water {n} atoms pbc  net force {float(net):.2e} Ha/A (sum |F| {float(fsum):.2e});  translated box: "
           f"|dE| = {dE:.2e} Ha, max|dF| = {dF:.2e} Ha/A")
    assert float(net) < 1e-8 * float(fsum)
    assert dE < 1e-9 * n and dF < F_REG
    del out, moved
    torch.cuda.empty_cache()


@pytest.mark.parametrize("elements", [(0, 1), (0, 3), (1, 2, 3), (0, 2, 6), (0, 1, 2, 3), (2, 4, 5, 6), (0, 1, 2, 3, 4, 5, 6)])
def test_large_systems_of_any_composition_against_the_oracle(dev, elements):
    """The large-system product path -- 256-row tiles, slab masks, skinny layer-0 backward with 2 .. 6+ flagged slabs (the
    4-slab case multiplied a stale register until round 3), species numbered present-ones-first, the spatial path -- on a
    17 496-atom periodic box whose atoms are given the listed elements at random: sampled atoms against the fp64 oracle on
    the clusters around them; shuffled and sharded evaluation agree with it."""
    from bench import water_box
    from oracle.sampled_parity import sampled_parity

    sp_np, x_np, cell_np = water_box(18)
    rs = np.random.RandomState(sum(elements))
    sp_np = rs.choice(np.asarray(elements), size=sp_np.shape).astype(sp_np.dtype)
    model = get_model("ani2x", 29, dev, neighborlist="cell", row_capacity=192)
    sp, x, cell = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    out = model.energies_and_forces(sp, x, cell, pbc, check_overflow=True)
    res = sampled_parity(sp, x, cell, out.atomic_energies, out.forces, seeded_state("ani2x", 8, 29), "ani2x", 8,
                         n_sample=24, seed=5)
    report(f"box   elements {elements} {sp.numel()} atoms: max|e_atom err| = {res['max_dE_atom']:.2e}  max|F err| = {res['max_dF']:.2e}")
    # (element soups on water coordinates have forces of tens of Ha/A: gates relative to the largest force)
    fmax = max(1.0, float(out.forces.abs().max()))
    assert res["max_dE_atom"] < E_ATOM_TOL * max(1.0, float(out.atomic_energies.abs().max())) and res["max_dF"] < F_TOL * fmax
    assert res["max_dE_atom"] <= E_ATOM_REG * max(1.0, float(out.atomic_energies.abs().max())) and res["max_dF"] <= F_REG * fmax
    e = torch.zeros(1, dtype=torch.float64, device=dev)
    f = torch.zeros_like(x)
    for rank in range(3):
        part = model.energies_and_forces(sp, x, cell, pbc, shard=(rank, 3))
        e += part.energies
        f += part.forces
    assert float((f - out.forces).abs().max()) < 5e-6 * fmax and abs(float(e - out.energies)) < 1e-7 * sp.numel() * fmax


def _canonical_pairs(idx, diff):
    """(i, j, diff) rows with i < j (or the lexicographically positive image of an atom with itself), sorted."""
    i, j, d = idx[0].copy(), idx[1].copy(), diff.astype(np.float64).copy()
    swap = i > j
    i[swap], j[swap] = idx[1][swap], idx[0][swap]
    d[swap] *= -1.0
    same = i == j
    first = np.where(np.abs(d[:, 0]) > 1e-6, d[:, 0], np.where(np.abs(d[:, 1]) > 1e-6, d[:, 1], d[:, 2]))
    d[same & (first < 0)] *= -1.0
    key = np.lexsort((np.round(d[:, 2], 3), np.round(d[:, 1], 3), np.round(d[:, 0], 3), j, i))
    return i[key], j[key], d[key]


@pytest.mark.parametrize("name", ["water_pbc_ani2x", "triclinic_pbc_ani2x", "small_ani2x"])
def test_cell_list_half_list_matches_reference(dev, name):
    """torchani_amd.aev.cell_list (anihip_nbr_build_cell + anihip_nbr_rows_to_half) against the reference's own half list
    of the fixture (tests/golden/nbrs_*.npz from gen_golden_nbrs.py): the same set of (i, j, image) pairs, distances and
    displacement vectors; and the list feeds compute_from_neighbors like the reference's."""
    from torchani_amd.aev import AEVComputer, cell_list
    from torchani_amd.weights import arch_spec

    g = load_golden(name)
    nb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"nbrs_{name}.npz"))
    consts = arch_spec(g["kind"])[1]
    sp, x, cell, pbc = to_dev(g, dev)
    assert sp.shape[0] == 1
    pbc_t = None if pbc is None else torch.tensor(pbc)
    got = cell_list(consts.Rcr, sp, x, cell, pbc_t)
    gi, gj, gd = _canonical_pairs(got.indices.cpu().numpy(), got.diff_vectors.cpu().numpy())
    ri, rj, rd = _canonical_pairs(nb["indices"], nb["diff_vectors"])
    assert len(gi) == len(ri), f"{len(gi)} pairs, the reference has {len(ri)}"
    assert np.array_equal(gi, ri) and np.array_equal(gj, rj)
    assert np.abs(gd - rd).max() < 5e-5
    assert np.abs(got.distances.cpu().numpy() - got.diff_vectors.norm(dim=1).cpu().numpy()).max() < 1e-6
    report(f"half  {name:22s} cell_list -> {len(gi)} pairs == reference's list, max|diff err| = {np.abs(gd - rd).max():.1e}")
    aevc = AEVComputer(consts, row_capacity=256).to(dev)
    aev = aevc.compute_from_neighbors(sp, x, got).cpu().numpy().reshape(sp.numel(), -1)
    assert np.abs(aev[g["aev_rows"]] - g["aev"]).max() < AEV_TOL
