"""Sampled parity of a LARGE periodic system against the fp64 oracle (test infrastructure: used by bench.py after its
timed loop and by tests/; never by the product path).

The per-atom energy E_i depends on the atoms within the radial cutoff Rcr of atom i, and the force on atom i,
F_i = -sum_j dE_j/dr_i, on the atoms within 2 Rcr (the E_j of the neighbors j, each with ITS neighbors).  So the cluster of
all atoms (minimum image) within 2 Rcr of a sampled atom, evaluated WITHOUT periodicity, reproduces that atom's E_i and F_i
of the infinite periodic system exactly -- which lets the CPU oracle check a 2.3 M-atom result atom by atom at the price of
~450 atoms per sample (reference: the whole path, torchani/grad.py:263-290 on arch.py:302-349; the reference's own
large-system check samples the same way, tools/scaling-aev-benchmark.py:262-277)."""
from __future__ import annotations

import typing as tp

import numpy as np


def cut_clusters(species, coords, cell, centers: np.ndarray, radius: float):
    """species [N] (element indices), coords [N, 3], cell [3, 3] (orthorhombic, every edge >= 2 radius) as torch tensors
    on any device; centers: atom indices.  Returns (cluster species [K, A] int64 padded with -1, coordinates [K, A, 3]
    float64 relative to the centre atom, which is atom 0 of its cluster)."""
    import torch

    box = torch.diagonal(cell).to(torch.float64)
    off = cell.to(torch.float64) - torch.diag(box)
    if float(off.abs().max()) > 1e-9 or float(box.min()) < 2.0 * radius:
        raise ValueError("sampled parity needs an orthorhombic cell with every edge >= 2 * radius")
    x = coords.reshape(-1, 3).to(torch.float64)
    sp = species.reshape(-1)
    out_s, out_x = [], []
    for c in centers.tolist():
        d = x - x[c]
        d = d - box * torch.round(d / box)
        r2 = (d * d).sum(dim=1)
        r2[c] = -1.0   # the centre first
        idx = torch.nonzero(r2 < radius * radius).reshape(-1)
        idx = idx[torch.argsort(r2[idx])]
        out_s.append(sp[idx].cpu().numpy().astype(np.int64))
        out_x.append(d[idx].cpu().numpy())
    A = max(len(s) for s in out_s)
    S = np.full((len(out_s), A), -1, dtype=np.int64)
    X = np.zeros((len(out_s), A, 3), dtype=np.float64)
    for k, (s, xx) in enumerate(zip(out_s, out_x)):
        S[k, :len(s)] = s
        X[k, :len(s)] = xx
    return S, X


def sampled_parity(species, coords, cell, atomic_energies, forces, state_dict: tp.Mapping[str, np.ndarray], kind: str = "ani2x",
                   n_members: int = 8, n_sample: int = 512, seed: int = 0, batch: int = 128,
                   candidates: tp.Optional[np.ndarray] = None, threads: tp.Optional[int] = None) -> tp.Dict[str, tp.Any]:
    """Compare ``atomic_energies`` [N] (NN part, ensemble mean) and ``forces`` [N, 3] of a periodic system with the fp64
    oracle on ``n_sample`` random real atoms -- of ``candidates`` (atom indices) when given: a rank of a sharded run holds
    the results of the atoms it OWNS, the clusters are cut from the whole box (which every rank has).
    ``threads``: OpenMP threads of the oracle for this call (a rank started by torch.distributed.run inherits
    OMP_NUM_THREADS=1: 128 clusters then take a minute on one core while the other ranks wait at a barrier).
    Returns {n, max_dE_atom, max_dF, cluster_atoms_mean, seconds, oracle_threads}."""
    import time

    import torch

    from oracle import oracle as orc
    from torchani_amd.weights import arch_spec

    symbols, consts, _ = arch_spec(kind)
    p = orc.params_2x() if kind == "ani2x" else orc.params_1x()
    dims, flat = orc.pack_networks(state_dict, symbols, n_members)
    o64 = orc.Oracle("f64")
    threads_before = o64.num_threads()
    if threads:
        o64.set_threads(int(threads))
    threads_used = o64.num_threads()
    sp = species.reshape(-1)
    real = torch.nonzero(sp >= 0).reshape(-1).cpu().numpy()
    if candidates is not None:
        real = np.intersect1d(real, np.asarray(candidates, dtype=np.int64))
    rs = np.random.RandomState(seed)
    centers = np.sort(rs.choice(real, size=min(n_sample, len(real)), replace=False))
    e_dev = atomic_energies.reshape(-1)[torch.from_numpy(centers).to(atomic_energies.device)].double().cpu().numpy()
    f_dev = forces.reshape(-1, 3)[torch.from_numpy(centers).to(forces.device)].double().cpu().numpy()
    t0 = time.perf_counter()
    max_de = max_df = 0.0
    sizes = []
    for b0 in range(0, len(centers), batch):
        S, X = cut_clusters(species, coords, cell, centers[b0:b0 + batch], 2.0 * consts.Rcr)
        sizes.append(float((S >= 0).sum(axis=1).mean()))
        ref = o64.energy_forces(p, S, X, dims, flat, n_members, sae=None)
        max_de = max(max_de, float(np.abs(ref["atomic_energies"][:, 0] - e_dev[b0:b0 + batch]).max()))
        max_df = max(max_df, float(np.abs(ref["forces"][:, 0] - f_dev[b0:b0 + batch]).max()))
    if threads:
        o64.set_threads(threads_before)
    return {"n": int(len(centers)), "oracle_threads": threads_used, "max_dE_atom": max_de, "max_dF": max_df, "cluster_atoms_mean": float(np.mean(sizes)),
            "radius_A": 2.0 * consts.Rcr, "oracle": "oracle/ani_oracle.c fp64, non-periodic clusters around the sampled atoms",
            "gate_dE_atom": 1e-5, "gate_dF": 1e-4,
            # (regression gates ~20x the measured error, as in tests/test_gpu_parity.py: a kernel bug inside the parity gates fails these)
            "regression_gate_dE_atom": 1e-6, "regression_gate_dF": 5e-6, "seconds": time.perf_counter() - t0}
