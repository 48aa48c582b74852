"""Shared helpers for the tests: golden fixtures, seeded weights, oracle packing."""
from __future__ import annotations

import functools
import glob
import os

import numpy as np

from oracle import oracle as orc
from torchani_amd.weights import random_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                      if not os.path.basename(p).startswith(("nbrs_", "wgrads_", "stress_", "fgrads_", "cfg3_", "pairs_", "pairs2_", "d3_", "x2r_", "x2rtrain_", "mbis_", "simple_", "hess_", "grid_")))   # reference neighbor lists /
#                                                                                      weight-gradient digests
WGRAD_NAMES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "wgrads_*.npz")))
STRESS_NAMES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "stress_*.npz")))
FGRAD_NAMES = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "fgrads_*.npz")))
WGRAD_BLOCK = 4096


def wgrad_upstream(C, A):
    """Per-atom loss weights of tests/golden/gen_golden_wgrads.py."""
    k = np.arange(C * A, dtype=np.float64)
    return (0.5 + np.modf(0.37 * k)[0]).reshape(C, A)


def wgrad_digest(flat):
    """Block sums / pattern dot products / sampled heads of a packed gradient vector (gen_golden_wgrads.py)."""
    n = flat.shape[0]
    nb = (n + WGRAD_BLOCK - 1) // WGRAD_BLOCK
    pad = np.zeros(nb * WGRAD_BLOCK, dtype=np.float64)
    pad[:n] = flat
    pat = np.cos(0.37 * np.arange(nb * WGRAD_BLOCK, dtype=np.float64))
    blocks = pad.reshape(nb, WGRAD_BLOCK)
    return blocks.sum(axis=1), (pad * pat).reshape(nb, WGRAD_BLOCK).sum(axis=1), blocks[::97, :64].copy()


def load_wgrads(base):
    with np.load(os.path.join(GOLDEN_DIR, "wgrads_" + base + ".npz")) as z:
        return {k: z[k] for k in z.files}


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["kind"] = str(g["kind"])
    g["seed"] = int(g["seed"])
    g["n_members"] = int(g["n_members"])
    g["symbols"] = [str(s) for s in g["symbols"]]
    g.setdefault("cell", None)
    g.setdefault("pbc", None)
    g["cutoff_fn"] = str(g["cutoff_fn"]) if "cutoff_fn" in g else "cosine"
    return g


@functools.lru_cache(maxsize=4)
def seeded_state(kind, n_members, seed):
    return random_state_dict(kind, n_members, seed)


@functools.lru_cache(maxsize=4)
def oracle_networks(kind, n_members, seed):
    from torchani_amd.weights import arch_spec

    symbols, _, _ = arch_spec(kind)
    sd = seeded_state(kind, n_members, seed)
    dims, flat = orc.pack_networks(sd, symbols, n_members)
    sae = sd["energy_shifter.self_energies"].astype(np.float64)
    return dims, flat, sae


def oracle_params(kind, cutoff_fn="cosine"):
    return orc.params_2x(cutoff_fn) if kind == "ani2x" else orc.params_1x(cutoff_fn)


def load_sampled(name):
    """Large-system fixtures of tests/golden/gen_golden_configs.py (inputs whole, outputs on a sample of atoms)."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["seed"] = int(g["seed"])
    g["species"] = g["species"].astype(np.int64)
    g.setdefault("cell", None)
    g.setdefault("pbc", None)
    return g


def load_stress(base):
    with np.load(os.path.join(GOLDEN_DIR, "stress_" + base + ".npz")) as z:
        return {k: z[k] for k in z.files}


def load_fgrads(base):
    with np.load(os.path.join(GOLDEN_DIR, "fgrads_" + base + ".npz")) as z:
        return {k: z[k] for k in z.files}


def fgrad_direction(species):
    """Coordinate-space direction of tests/golden/gen_golden_fgrads.py (zero on padding atoms)."""
    C, A = species.shape
    q = 3.0 * np.arange(C * A, dtype=np.float64)
    t = np.stack([np.modf(0.37 * q)[0], np.modf(0.61 * q)[0] - 0.5, 0.25 - np.modf(0.13 * q)[0]], axis=-1)
    return t.reshape(C, A, 3) * (species >= 0)[..., None]


def import_reference():
    """Make ``import torchani`` find the reference tree (build container only): stub modules for its optional imports that
    are not installed here (h5py, zarr).  CPU tests only -- nothing on the GPU box may depend on it."""
    import sys
    import types

    class _Any:
        def __class_getitem__(cls, k):
            return cls

    for name in ("h5py", "zarr"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k in ("File", "Group", "Dataset", "Datatype"):
                setattr(m, k, type(k, (_Any,), {}))
            sys.modules[name] = m
    os.environ["TORCHANI_NO_WARN_EXTENSIONS"] = "1"
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
