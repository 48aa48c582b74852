"""Utilities for systems with atomic charges (torchani/electro.py): charge normalizers for ANIq models and dipoles from
atomic charges.  Pure tensor code on whatever device the inputs live on; the charges themselves come from the charge
networks of torchani_amd.models.ANIq (the fused network kernel)."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

__all__ = ["BaseChargeNormalizer", "ChargeNormalizer", "DipoleComputer", "compute_dipole"]

# resources/atomic_constants.json "mass" (amu) by atomic number, the elements the engine supports (others: pass ``masses``)
ATOMIC_MASS_BY_Z: tp.Dict[int, float] = {1: 1.008, 6: 12.011, 7: 14.007, 8: 15.999, 9: 18.99840316, 16: 32.06, 17: 35.45}


class BaseChargeNormalizer(torch.nn.Module):
    """Leaves the raw charges as they are (electro.py:22-26: ``simple_aniq(normalize=False)``)."""

    def forward(self, elem_idxs: Tensor, raw_charges: Tensor, charge: int = 0) -> Tensor:
        return raw_charges


class ChargeNormalizer(BaseChargeNormalizer):
    """Shift raw atomic charges so that they add up to the total charge (electro.py:29-87): the excess is distributed
    with per-element weights, optionally scaled by the squared raw charges."""

    def __init__(self, symbols: tp.Sequence[str], weights: tp.Sequence[float] = (),
                 scale_weights_by_charges_squared: bool = False) -> None:
        super().__init__()
        if not weights:
            weights = [1.0] * len(symbols)
        self.register_buffer("weights", torch.tensor(list(weights), dtype=torch.float), persistent=False)
        self.scale_weights_by_charges_squared = scale_weights_by_charges_squared

    @classmethod
    def from_electronegativity_and_hardness(cls, symbols: tp.Sequence[str], electronegativity: tp.Sequence[float] = (),
                                            hardness: tp.Sequence[float] = (),
                                            scale_weights_by_charges_squared: bool = False) -> "ChargeNormalizer":
        from ..constants import ELECTRONEGATIVITY_HARDNESS as EH

        en = list(electronegativity) if electronegativity else [EH[s][0] for s in symbols]
        hd = list(hardness) if hardness else [EH[s][1] for s in symbols]
        return cls(symbols, [(e / h) ** 2 for e, h in zip(en, hd)], scale_weights_by_charges_squared)

    def factor(self, elem_idxs: Tensor, raw_charges: Tensor) -> Tensor:
        w = self.weights.to(raw_charges.dtype)[elem_idxs.clamp(min=0)].masked_fill(elem_idxs == -1, 0.0)
        if self.scale_weights_by_charges_squared:
            w = w * raw_charges ** 2
        return w / torch.sum(w, dim=-1, keepdim=True)

    def forward(self, elem_idxs: Tensor, raw_charges: Tensor, charge: int = 0) -> Tensor:
        excess = charge - raw_charges.sum(dim=-1, keepdim=True)
        return raw_charges + excess * self.factor(elem_idxs, raw_charges)


class DipoleComputer(torch.nn.Module):
    """Dipoles in e A from atomic charges (electro.py:96-158): sum_i q_i (r_i - r_ref) with r_ref the center of mass,
    the center of geometry, or the origin; padding atoms (-1) do not count."""

    def __init__(self, masses: tp.Iterable[float] = (), reference: str = "center_of_mass", device=None, dtype=None) -> None:
        super().__init__()
        if reference not in ("center_of_mass", "center_of_geometry", "origin"):
            raise ValueError(f"Unknown reference {reference!r}")
        masses = list(masses)
        if not masses:   # indexed by atomic number, like the reference's MASS tuple
            masses = [0.0] * (max(ATOMIC_MASS_BY_Z) + 1)
            for z, m in ATOMIC_MASS_BY_Z.items():
                masses[z] = m
        self.register_buffer("atomic_masses", torch.tensor(masses, device=device, dtype=dtype), persistent=False)
        self._center_of_mass = reference == "center_of_mass"
        self._skip = reference == "origin"

    def forward(self, atomic_nums: Tensor, coordinates: Tensor, charges: Tensor) -> Tensor:
        assert atomic_nums.shape == charges.shape == coordinates.shape[:-1]
        return torch.sum(charges.unsqueeze(-1) * self._displace_to_reference(atomic_nums, coordinates), dim=1)

    def _displace_to_reference(self, species: Tensor, coordinates: Tensor) -> Tensor:
        if self._skip:
            return coordinates
        mask = species == -1
        if self._center_of_mass:
            assert not (species == 0).any(), "Input should be atomic numbers"
            known = species < self.atomic_masses.numel()
            w = self.atomic_masses.to(coordinates.dtype)[species.clamp(min=0) * known].masked_fill(mask, 0.0)
            if (~known).any() or ((w == 0) & ~mask).any():
                raise ValueError("no mass for some of the atomic numbers: pass masses=")
        else:
            w = (~mask).to(coordinates.dtype)
        w = (w / w.sum(dim=1, keepdim=True)).unsqueeze(-1)
        centered = coordinates - (coordinates * w).sum(dim=1, keepdim=True)
        return centered.masked_fill(mask.unsqueeze(-1), 0.0)


def compute_dipole(species: Tensor, coordinates: Tensor, charges: Tensor, reference: str = "center_of_mass") -> Tensor:
    """Convenience wrapper over DipoleComputer (electro.py:161-180)."""
    return DipoleComputer(reference=reference, device=species.device, dtype=coordinates.dtype)(species, coordinates, charges)
