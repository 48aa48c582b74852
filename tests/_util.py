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
                      if not os.path.basename(p).startswith("nbrs_"))   # nbrs_*: reference neighbor lists


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
