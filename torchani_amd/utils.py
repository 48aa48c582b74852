"""Host-side helpers with the reference's names (torchani/utils.py): the few that callers of the hot path use.

These run on whatever device the tensors live on; they are plumbing around the engine, not part of it.
"""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from .constants import ATOMIC_NUMBER, PADDING_SPECIES, SYMBOLS_1X, SYMBOLS_2X, linspace  # noqa: F401

# utils.py:67-74: padding values per property
PADDING: tp.Dict[str, float] = {"species": PADDING_SPECIES, "numbers": PADDING_SPECIES, "atomic_numbers": PADDING_SPECIES,
                                "coordinates": 0.0, "forces": 0.0, "energies": 0.0}
ATOMIC_KEYS = ("species", "numbers", "atomic_numbers", "coordinates", "forces")


def cumsum_from_zero(input_: Tensor) -> Tensor:
    """Exclusive cumulative sum along dim 0 (utils.py:132-136)."""
    out = torch.zeros_like(input_)
    if input_.shape[0] > 1:
        out[1:] = torch.cumsum(input_[:-1], dim=0)
    return out


def nonzero_in_chunks(tensor: Tensor, chunk_size: int = 2**31 - 1) -> Tensor:
    """Flat indices of the non-zero elements, evaluated chunk-wise for tensors beyond INT_MAX (utils.py:139-162)."""
    flat = tensor.view(-1)
    if flat.numel() <= chunk_size:
        return flat.nonzero().view(-1)
    parts = [flat[o:o + chunk_size].nonzero().view(-1) + o for o in range(0, flat.numel(), chunk_size)]
    return torch.cat(parts)


def fast_masked_select(x: Tensor, mask: Tensor, idx: int) -> Tensor:
    """x.index_select(idx, nonzero(mask)): masked_select along one dimension (utils.py:165-171)."""
    return x.index_select(idx, nonzero_in_chunks(mask))


def pad_atomic_properties(properties: tp.Sequence[tp.Mapping[str, Tensor]],
                          padding_values: tp.Optional[tp.Dict[str, float]] = None) -> tp.Dict[str, Tensor]:
    """[{'species': [c1, a1], 'coordinates': [c1, a1, 3], 'energies': [c1]}, ...] -> one dictionary of tensors padded
    along the atom dimension to the largest molecule (species with -1, utils.py:174-221)."""
    pad = PADDING if padding_values is None else padding_values
    first = properties[0]
    per_atom = [k for k, v in first.items() if v.dim() > 1]
    per_mol = [k for k, v in first.items() if v.dim() == 1]
    counts = [p[per_atom[0]].shape[0] for p in properties]
    out: tp.Dict[str, Tensor] = {k: torch.cat([p[k] for p in properties]) for k in per_mol}
    for k in per_atom:
        ref = first[k]
        dtype = torch.long if ref.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32) else ref.dtype
        shape = [sum(counts), max(p[k].shape[1] for p in properties)] + list(ref.shape[2:])
        buf = torch.full(shape, pad.get(k, 0.0), dtype=dtype, device=ref.device)
        row = 0
        for n, p in zip(counts, properties):
            buf[row:row + n, :p[k].shape[1]] = p[k]
            row += n
        out[k] = buf
    return out


def strip_redundant_padding(properties: tp.Dict[str, Tensor],
                            atomic_properties: tp.Iterable[str] = ATOMIC_KEYS) -> tp.Dict[str, Tensor]:
    """Drop atom columns that are padding in every molecule (utils.py:224-234)."""
    keep = (properties["species"] >= 0).any(dim=0).nonzero().view(-1)
    for k in atomic_properties:
        if k in properties:
            properties[k] = properties[k].index_select(1, keep)
    return properties


def map_to_central(coordinates: Tensor, cell: Tensor, pbc: Tensor) -> Tensor:
    """Wrap atoms into the unit cell along the periodic lattice vectors (utils.py:237-255); the neighbor builders of
    the engine do this themselves (fp64 per atom), so calling it first is never required."""
    frac = coordinates @ torch.inverse(cell)
    frac = frac - frac.floor() * pbc.to(frac.dtype)
    return frac @ cell
