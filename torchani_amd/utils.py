"""Host-side helpers with the reference's names (torchani/utils.py): the few that callers of the hot path use.

These run on whatever device the tensors live on; they are plumbing around the engine, not part of it.
"""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from .constants import ATOMIC_NUMBER, PADDING_SPECIES, SYMBOLS_1X, SYMBOLS_2X, SYMBOLS_2X_ZNUM_ORDER, linspace  # noqa: F401

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


# ---- symbol / number / mass converters (utils.py:257-473) ------------------------------------------------------------
class _NumbersToSymbols(torch.nn.Module):
    def __init__(self, table: tp.Mapping[int, str]) -> None:
        super().__init__()
        self.symbol_dict = dict(table)

    def forward(self, species: Tensor) -> tp.List[str]:
        """1-D tensor of numbers -> list of symbols, padding (-1) left out."""
        assert species.dim() == 1, "Only 1D tensors supported"
        return [self.symbol_dict[int(x)] for x in species.tolist() if x != -1]

    def __len__(self) -> int:
        return len(self.symbol_dict)


class _SymbolsToNumbers(torch.nn.Module):
    def __init__(self, table: tp.Mapping[str, int], device=None) -> None:
        super().__init__()
        self.symbol_dict = dict(table)
        self.register_buffer("_dummy", torch.empty(0, device=device), persistent=False)

    def forward(self, species: tp.Sequence[str]) -> Tensor:
        """Sequence of chemical symbols -> int64 tensor (on the module's device)."""
        return torch.tensor([self.symbol_dict[x] for x in species], dtype=torch.long, device=self._dummy.device)

    def __len__(self) -> int:
        return len(self.symbol_dict)


class AtomicNumbersToChemicalSymbols(_NumbersToSymbols):
    """tensor([6, 1, 1, 1]) -> ['C', 'H', 'H', 'H'] (utils.py:277-305)."""

    def __init__(self) -> None:
        from .extras.io import PERIODIC_TABLE

        super().__init__({z: s for z, s in enumerate(PERIODIC_TABLE) if s})


class IntsToChemicalSymbols(_NumbersToSymbols):
    """Element indices of a model -> symbols: IntsToChemicalSymbols(['H', 'C', 'N', 'O'])(tensor([3, 0, 0, -1])) ->
    ['O', 'H', 'H'] (utils.py:308-333)."""

    def __init__(self, symbols: tp.Sequence[str]) -> None:
        if isinstance(symbols, str):
            raise ValueError("symbols must be a sequence of str, but it can't be a str")
        super().__init__(dict(enumerate(symbols)))


class ChemicalSymbolsToAtomicNumbers(_SymbolsToNumbers):
    """['C', 'S', 'O'] -> tensor([6, 16, 8]) (utils.py:356-373)."""

    def __init__(self, device=None) -> None:
        from .extras.io import PERIODIC_TABLE

        super().__init__({s: z for z, s in enumerate(PERIODIC_TABLE) if s}, device=device)


class ChemicalSymbolsToInts(_SymbolsToNumbers):
    """Symbols -> element indices of a model built for ``symbols`` (utils.py:376-403)."""

    def __init__(self, symbols: tp.Sequence[str], device=None) -> None:
        if isinstance(symbols, str):
            raise ValueError("symbols must be a sequence of str, but it can't be a str")
        super().__init__({s: i for i, s in enumerate(symbols)}, device=device)


class AtomicNumbersToMasses(torch.nn.Module):
    """Atomic numbers -> masses in amu, padding -> 0 (utils.py:406-439).  ``masses`` is indexed by atomic number; the default
    table covers the elements the engine supports."""

    def __init__(self, masses: tp.Iterable[float] = (), device=None, dtype=None) -> None:
        super().__init__()
        masses = list(masses)
        if not masses:
            from .extras.electro import ATOMIC_MASS_BY_Z

            masses = [0.0] * (max(ATOMIC_MASS_BY_Z) + 1)
            for z, m in ATOMIC_MASS_BY_Z.items():
                masses[z] = m
        self.register_buffer("atomic_masses", torch.tensor(masses, device=device, dtype=dtype), persistent=False)

    def forward(self, atomic_numbers: Tensor) -> Tensor:
        assert not (atomic_numbers == 0).any(), "Input should be atomic numbers"
        mask = atomic_numbers == -1
        known = atomic_numbers < self.atomic_masses.numel()
        m = self.atomic_masses[atomic_numbers.clamp(min=0) * known].masked_fill(mask, 0.0)
        if (~known).any() or ((m == 0) & ~mask).any():
            raise ValueError("no mass for some of the atomic numbers: pass masses=")
        return m


def atomic_numbers_to_masses(atomic_numbers: Tensor, dtype: torch.dtype = torch.float) -> Tensor:
    """Convenience wrapper over AtomicNumbersToMasses (utils.py:442-456)."""
    return AtomicNumbersToMasses(device=atomic_numbers.device, dtype=dtype)(atomic_numbers)


get_atomic_masses = atomic_numbers_to_masses   # (the reference's older name)


def sort_by_atomic_num(it: tp.Iterable[str]) -> tp.Tuple[str, ...]:
    """Chemical symbols sorted by atomic number (utils.py:463-473)."""
    from .extras.io import PERIODIC_TABLE

    if isinstance(it, str):
        it = (it,)
    return tuple(sorted(it, key=PERIODIC_TABLE.index))
