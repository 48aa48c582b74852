"""Deterministic random ANI network parameters under the reference's state-dict key names.

The published ANI-2x parameters cannot be fetched offline (the reference downloads them,
arch.py:1185-1220), so tests and bench.py use the ANI architecture with seeded random weights.  The
generator is numpy's frozen legacy ``RandomState`` so the very same numbers are reproduced on any box
(the golden fixtures in tests/golden were computed by the reference with exactly these weights).
Keys follow SURVEY section 5: ``potentials.nnp.neural_networks.members.{m}.atomics.{Sym}.layers.{l}.weight``
etc.; a real ``ani2x_state_dict.pt`` uses the same names and loads through the same code path.
"""
from __future__ import annotations

import typing as tp

import numpy as np

from .constants import (GSAES_B973C_DEF2MTZVP, GSAES_CCSDT_STAR_CBS, GSAES_R2SCAN3C, GSAES_WB97X_631GD, HIDDEN_DIMS_1X, HIDDEN_DIMS_2X, SYMBOLS_1X, SYMBOLS_2X,
                        SYMBOLS_2X_ZNUM_ORDER, aev_constants_1x, aev_constants_2x, aev_constants_simple)

NN_PREFIX = "potentials.nnp.neural_networks."


def arch_spec(kind: str):
    """(symbols, AEVConstants, hidden dims) of a builtin architecture (models.py:112-119,185-193)."""
    if kind == "ani2x":
        return SYMBOLS_2X, aev_constants_2x(), HIDDEN_DIMS_2X
    if kind == "ani1x":
        return SYMBOLS_1X, aev_constants_1x(), HIDDEN_DIMS_1X
    if kind == "ani1ccx":   # models.py:128-162: the ANI-1x architecture (trained to CCSD(T)*/CBS)
        return SYMBOLS_1X, aev_constants_1x(), HIDDEN_DIMS_1X
    if kind in ("ani2xr", "ani2dr"):   # models.py:252-320: simple_ani with the ANI-2x widths, GELU, no biases
        return SYMBOLS_2X_ZNUM_ORDER, aev_constants_simple(), HIDDEN_DIMS_2X
    if kind.startswith("anir2s"):   # models.py:325-368: the ANI-2x AEV with the smooth envelope, GELU, no biases
        return SYMBOLS_2X, aev_constants_2x(cutoff_fn="smooth"), HIDDEN_DIMS_2X
    raise ValueError(f"Unknown architecture {kind!r}")


def arch_networks(kind: str) -> tp.Tuple[str, bool]:
    """(activation, bias) of the atomic networks of a builtin architecture (arch.py:1010-1011 for the -r models)."""
    return ("gelu", False) if kind in ("ani2xr", "ani2dr") or kind.startswith("anir2s") else ("celu", True)


def arch_gsaes(kind: str) -> tp.Dict[str, float]:
    """Self energies (ground-state atomic energies of the model's level of theory, arch.py:1053 / models.py)."""
    if kind == "ani2dr":
        return GSAES_B973C_DEF2MTZVP
    if kind == "ani1ccx":
        return GSAES_CCSDT_STAR_CBS
    if kind.startswith("anir2s"):   # "anir2s", "anir2s_water", ...
        return GSAES_R2SCAN3C[kind[7:] or None]
    return GSAES_WB97X_631GD


def random_state_dict(kind: str = "ani2x", n_members: int = 8, seed: int = 0,
                      scale: float = 1.0) -> tp.Dict[str, np.ndarray]:
    """fp32 parameters, uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch.nn.Linear's default."""
    symbols, consts, hidden = arch_spec(kind)
    rs = np.random.RandomState(seed)
    out: tp.Dict[str, np.ndarray] = {}
    for m in range(n_members):
        for sym in symbols:
            dims = (consts.out_dim,) + tuple(hidden[sym]) + (1,)
            nl = len(dims) - 1
            for l in range(nl):
                name = f"layers.{l}" if l < nl - 1 else "final_layer"
                bound = scale / np.sqrt(dims[l])
                base = f"{NN_PREFIX}members.{m}.atomics.{sym}.{name}."
                out[base + "weight"] = rs.uniform(-bound, bound, (dims[l + 1], dims[l])).astype(np.float32)
                if arch_networks(kind)[1]:
                    out[base + "bias"] = rs.uniform(-bound, bound, (dims[l + 1],)).astype(np.float32)
    out["energy_shifter.self_energies"] = np.asarray(
        [arch_gsaes(kind)[s] for s in symbols], dtype=np.float32)
    return out


def random_charge_state_dict(seed: int = 0) -> tp.Dict[str, np.ndarray]:
    """Seeded parameters of the ANI-mbis charge networks (ANI-2x widths, two outputs, no biases) under the key names of
    the reference's charge_nn_state_dict.pt (``atomics.{Sym}.layers.{l}.weight`` / ``final_layer.weight``)."""
    symbols, consts, hidden = arch_spec("ani2x")
    rs = np.random.RandomState(1000 + seed)
    out: tp.Dict[str, np.ndarray] = {}
    for sym in symbols:
        dims = (consts.out_dim,) + tuple(hidden[sym]) + (2,)
        nl = len(dims) - 1
        for l in range(nl):
            name = f"layers.{l}" if l < nl - 1 else "final_layer"
            bound = 3.0 / np.sqrt(dims[l])   # (bias-free GELU layers shrink the signal: keeps the charges ~0.1 e)
            out[f"atomics.{sym}.{name}.weight"] = rs.uniform(-bound, bound, (dims[l + 1], dims[l])).astype(np.float32)
    return out


def random_network_state_dict(symbols: tp.Sequence[str], in_dim: int, hidden: tp.Mapping[str, tp.Sequence[int]],
                              n_members: int = 1, seed: int = 0, bias: bool = True, out_dim: int = 1,
                              scale: float = 1.0) -> tp.Dict[str, np.ndarray]:
    """Seeded fp32 parameters (uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch.nn.Linear's default) for ANINetworks of
    any supported shape, under the reference's key names; one member: no ``members.{m}.`` level (arch.py:967-975)."""
    rs = np.random.RandomState(seed)
    out: tp.Dict[str, np.ndarray] = {}
    for m in range(n_members):
        for sym in symbols:
            dims = (in_dim,) + tuple(hidden[sym]) + (out_dim,)
            nl = len(dims) - 1
            for l in range(nl):
                name = f"layers.{l}" if l < nl - 1 else "final_layer"
                bound = scale / np.sqrt(dims[l])
                base = f"{NN_PREFIX}{f'members.{m}.' if n_members > 1 else ''}atomics.{sym}.{name}."
                out[base + "weight"] = rs.uniform(-bound, bound, (dims[l + 1], dims[l])).astype(np.float32)
                if bias:
                    out[base + "bias"] = rs.uniform(-bound, bound, (dims[l + 1],)).astype(np.float32)
    return out
