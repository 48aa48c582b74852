"""Pin the CPU oracle (oracle/ani_oracle.c) against fixtures produced by the reference itself.

The fixtures in tests/golden were written by tests/golden/gen_golden.py, which imports the reference
(fp64, pyaev) -- so agreement here to ~1e-12 means the C restatement IS the reference's arithmetic.
"""
import os

import numpy as np
import pytest

from _util import (FGRAD_NAMES, GOLDEN_NAMES, STRESS_NAMES, WGRAD_NAMES, fgrad_direction, load_fgrads, load_golden,
                   load_stress, load_wgrads, oracle_networks, oracle_params, wgrad_digest, wgrad_upstream)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
@pytest.mark.parametrize("cell_list", [False, True])
def test_oracle_matches_reference(oracle64, name, cell_list):
    g = load_golden(name)
    if cell_list and g["species"].shape[0] != 1:
        pytest.skip("cell list handles one system at a time (neighbors.py:373-381)")
    dims, flat, sae = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    out = oracle64.energy_forces(p, g["species"], g["coords"].astype(np.float64), dims, flat,
                                 g["n_members"], sae=sae, cell=g["cell"], pbc=g["pbc"],
                                 cell_list=cell_list, want_aev=True)
    C, A = g["species"].shape
    aev = out["aev"].reshape(C * A, -1)[g["aev_rows"]]
    assert np.abs(aev - g["aev"]).max() < 2e-13
    assert np.abs(out["atomic_energies"] - g["atomic_energies"]).max() < 1e-12
    assert np.abs(out["energies"] - g["energies"]).max() < 1e-9 * max(1.0, np.abs(g["energies"]).max())
    # Forces go through torch's fp64 CELU backward, which evaluates exp(x * (1 / float32(alpha))) instead
    # of exp(x / alpha) (a torch artefact, relative error up to 1.5e-8 * |x| / alpha); the oracle uses the
    # exact derivative, so whole-path forces agree to ~1e-9 while the AEV backward alone (below) agrees
    # to 1e-13.
    assert np.abs(out["forces"] - g["forces"]).max() < 2e-9
    w = np.random.RandomState(g["seed"] + 1000).uniform(-1.0, 1.0, out["aev"].shape)
    _, vjp = oracle64.aev(p, g["species"], g["coords"].astype(np.float64), g["cell"], g["pbc"],
                          cell_list=cell_list, grad_aev=w)
    assert np.abs(vjp - g["aev_vjp"]).max() < 1e-12 * max(1.0, np.abs(g["aev_vjp"]).max())
    # padding atoms: exactly zero everywhere (neighbors.py:72-82, nn/_containers.py:412-416)
    pad = g["species"] < 0
    assert np.all(out["forces"][pad] == 0) and np.all(out["atomic_energies"][pad] == 0)


@pytest.mark.parametrize("base", WGRAD_NAMES)
def test_oracle_weight_gradients_match_reference_autograd(oracle64, base):
    """d Loss / d (weights, biases) of the oracle against the digest of the reference's autograd gradients
    (tests/golden/gen_golden_wgrads.py).  Tolerance: torch's fp64 CELU backward artefact (see above), ~1e-8 relative."""
    g, w = load_golden(base), load_wgrads(base)
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    aev = oracle64.aev(p, g["species"], g["coords"].astype(np.float64), g["cell"], g["pbc"])
    up = wgrad_upstream(*g["species"].shape)
    grads = oracle64.mlp_weight_grads(g["species"], aev, up, dims, flat, n_members=g["n_members"])
    assert grads.shape[0] == int(w["n_params"])
    ae, _, _ = oracle64.mlp(g["species"], aev, dims, flat, n_members=g["n_members"], want_grad=False)
    assert abs(float((ae.reshape(up.shape) * up).sum()) - float(w["loss"])) < 1e-11
    sums, dots, heads = wgrad_digest(grads)
    scale = float(w["grad_abs_max"])
    assert abs(np.abs(grads).max() - scale) < 1e-7 * scale
    assert abs(np.linalg.norm(grads) - float(w["grad_l2"])) < 1e-7 * float(w["grad_l2"])
    assert np.abs(sums - w["block_sums"]).max() < 1e-7 * scale
    assert np.abs(dots - w["block_dots"]).max() < 1e-7 * scale
    assert np.abs(heads - w["block_heads"]).max() < 1e-7 * scale


@pytest.mark.parametrize("base", STRESS_NAMES)
@pytest.mark.parametrize("cell_list", [False, True])
def test_oracle_virial_matches_reference_scaling_stress(oracle64, base, cell_list):
    """sum_ij dE/d d_ij (x) d_ij of the oracle (the reference's fdotr virial, ase.py:164-168) against the reference's
    autograd derivative with respect to a strain of coordinates and cell (ase.py:170-173), tests/golden/gen_golden_stress.py."""
    g, st = load_golden(base), load_stress(base)
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    vir = oracle64.virial(p, g["species"], g["coords"].astype(np.float64), dims, flat, g["n_members"], g["cell"],
                          g["pbc"], cell_list=cell_list)
    # (torch's fp64 CELU backward artefact, see above: ~1e-8 relative)
    assert np.abs(vir - st["virial"]).max() < 2e-8 * max(1.0, np.abs(st["virial"]).max())
    assert np.abs(vir - vir.T).max() < 1e-12   # rotational invariance of the energy


@pytest.mark.parametrize("base", FGRAD_NAMES)
def test_oracle_force_training_gradients_match_reference(oracle64, base):
    """Second order: J t (the reference's AEV forward-mode derivative = cuaev double backward) and d(t . F)/d params
    against the reference's create_graph autograd (tests/golden/gen_golden_fgrads.py)."""
    g, f = load_golden(base), load_fgrads(base)
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    t = fgrad_direction(g["species"])
    aev, jt = oracle64.aev_jvp(p, g["species"], g["coords"].astype(np.float64), t, g["cell"], g["pbc"])
    C, A = g["species"].shape
    assert np.abs(jt.reshape(C * A, -1)[g["aev_rows"]] - f["aev_jvp"]).max() < 1e-11 * max(1.0, np.abs(f["aev_jvp"]).max())
    # Loss = t . F = sum_i (-J t)_i . dE/d aev_i
    val, grads = oracle64.mlp_tangent_weight_grads(g["species"], aev, -jt, dims, flat, n_members=g["n_members"])
    assert abs(val - float(f["loss"])) < 1e-9 * max(1.0, abs(float(f["loss"])))
    sums, dots, heads = wgrad_digest(grads)
    scale = float(f["grad_abs_max"])
    assert grads.shape[0] == int(f["n_params"])
    assert abs(np.abs(grads).max() - scale) < 1e-6 * scale
    assert np.abs(sums - f["block_sums"]).max() < 1e-6 * scale
    assert np.abs(dots - f["block_dots"]).max() < 1e-6 * scale
    assert np.abs(heads - f["block_heads"]).max() < 1e-6 * scale


def test_member_energies(oracle64):
    g = load_golden("simple2_ani2x")
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    aev = oracle64.aev(p, g["species"], g["coords"].astype(np.float64))
    ae, _, me = oracle64.mlp(g["species"], aev, dims, flat, n_members=8, want_members=True)
    assert np.abs(me.reshape(8, *g["species"].shape) - g["member_atomic_energies"]).max() < 1e-12
    # Ensemble == mean of members (tests/test_ensemble.py:20-36)
    assert np.abs(me.mean(0) - ae).max() < 1e-14


@pytest.mark.parametrize("name", ["water_pbc_ani2x", "triclinic_pbc_ani2x", "1hz5_ani2x", "dense90_ani2x"])
def test_cell_list_equals_brute_force(oracle64, name):
    g = load_golden(name)
    x = g["coords"].astype(np.float64)
    a = oracle64.neighbors(g["species"], x, 5.1, g["cell"], g["pbc"], cell_list=False)
    b = oracle64.neighbors(g["species"], x, 5.1, g["cell"], g["pbc"], cell_list=True)
    assert np.array_equal(a[0], b[0])  # same count per central atom
    for i in range(len(a[0]) - 1):
        sa, sb = slice(a[0][i], a[0][i + 1]), slice(b[0][i], b[0][i + 1])
        ka = np.lexsort((a[2][sa, 2], a[2][sa, 1], a[2][sa, 0], a[1][sa]))
        kb = np.lexsort((b[2][sb, 2], b[2][sb, 1], b[2][sb, 0], b[1][sb]))
        assert np.array_equal(a[1][sa][ka], b[1][sb][kb])
        assert np.allclose(a[2][sa][ka], b[2][sb][kb], atol=1e-12)


def test_f32_port_close_to_f64(oracle64):
    """The float build (bench.py's cpu_baseline 'port') is the same code; sanity-check it."""
    from oracle.oracle import Oracle

    o32 = Oracle("f32")
    g = load_golden("small_ani2x")
    dims, flat, sae = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    out = o32.energy_forces(p, g["species"], g["coords"], dims, flat, 8, sae=sae)
    assert np.abs(out["atomic_energies"] - g["atomic_energies"]).max() < 2e-5
    assert np.abs(out["forces"] - g["forces"]).max() < 2e-5


GRID_CASES = ["r8_a4z4_batch", "r24_a10z8_pbc", "r5_a3z5_dense"]


def load_grid(name):
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"grid_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", GRID_CASES)
def test_oracle_on_general_grids(oracle64, name):
    """Symmetry-function grids other than the published ones (AEVComputer.from_constants, aev/_computer.py:602-666; the
    templated cuAEV kernels, csrc/aev.cu:1687-1777): the oracle against the reference's own AEVs and vector-Jacobian
    products (tests/golden/gen_golden_grids.py) -- this pins the checker of the general HIP kernels."""
    from oracle import oracle as orc

    g = load_grid(name)
    p = orc.make_params(int(g["num_species"]), float(g["Rcr"]), float(g["Rca"]), float(g["EtaR"]), float(g["EtaA"]),
                        float(g["Zeta"]), g["ShfR"].tolist(), g["ShfA"].tolist(), g["ShfZ"].tolist(), str(g["cutoff_fn"]))
    aev, vjp = oracle64.aev(p, g["species"], g["coords"], g.get("cell"), g.get("pbc"), grad_aev=g["cotangent"].astype(np.float64))
    assert np.abs(aev - g["aev"]).max() < 2e-12 * max(1.0, np.abs(g["aev"]).max())
    assert np.abs(vjp - g["aev_vjp"]).max() < 2e-11 * max(1.0, np.abs(g["aev_vjp"]).max())
    # the forward-mode derivative J t (the reference's torch.autograd.functional.jvp of the same computer; the role of the
    # cuaev double backward, csrc/aev.cu:1986-2015): pins the checker of the general kernels' JVP instantiation
    from _util import fgrad_direction
    t = fgrad_direction(g["species"])
    _, jt = oracle64.aev_jvp(p, g["species"], g["coords"].astype(np.float64), t, g.get("cell"), g.get("pbc"))
    assert np.abs(jt.reshape(g["aev_jvp"].shape) - g["aev_jvp"]).max() < 2e-11 * max(1.0, np.abs(g["aev_jvp"]).max())


def test_sampled_parity_clusters_reproduce_the_periodic_oracle(oracle64):
    """oracle/sampled_parity.py, the check behind bench.py's `parity_sample` and the large-system tests, against the oracle
    itself: the non-periodic cluster within 2 Rcr of an atom gives that atom's energy and force of the periodic box -- and
    the call runs on the OpenMP threads it is given (a rank under torch.distributed.run inherits OMP_NUM_THREADS=1) and
    leaves the oracle's thread count as it found it."""
    import torch

    from bench import water_box
    from oracle.sampled_parity import sampled_parity
    from _util import seeded_state

    sp_np, x_np, cell_np = water_box(7)   # 1029 atoms, 21.7 A edges (>= 2 x 10.2 A)
    dims, flat, _ = oracle_networks("ani2x", 8, 3)
    full = oracle64.energy_forces(oracle_params("ani2x"), sp_np, x_np.astype(np.float64), dims, flat, 8, sae=None,
                                  cell=cell_np, pbc=(True, True, True), cell_list=True)
    sp, x, cell = torch.from_numpy(sp_np), torch.from_numpy(x_np), torch.from_numpy(cell_np)
    e = torch.from_numpy(full["atomic_energies"])
    f = torch.from_numpy(full["forces"])
    before = oracle64.num_threads()
    own = np.arange(0, sp_np.size, 3)   # (a rank's owned atoms: here the oxygens)
    res = sampled_parity(sp, x, cell, e, f, seeded_state("ani2x", 8, 3), "ani2x", 8, n_sample=6, seed=1, candidates=own,
                         threads=2)
    assert res["n"] == 6 and res["oracle_threads"] == 2 and oracle64.num_threads() == before
    assert res["max_dE_atom"] < 1e-11 and res["max_dF"] < 1e-10, res
    assert 300 < res["cluster_atoms_mean"] < 600
