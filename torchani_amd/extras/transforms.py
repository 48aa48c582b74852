"""Composable transforms of property batches (dictionaries of tensors: "species" atomic numbers [C, A], "coordinates",
"energies", "forces" ...) under the reference's names (torchani/transforms.py): what a training loop applies to a batch
before it reaches the model -- subtract the self energies, turn atomic numbers into the model's element indices, chain
such steps.  Pure tensor code on the batch's device.

Not here: ``SubtractRepulsionXTB`` / ``SubtractTwoBodyDispersionD3`` / ``SubtractEnergyAndForce`` (transforms.py:81-151)
evaluate a pair potential on its own; this package's pair potentials run on a model's neighbor rows
(``ANI.add_pair_potential``), so the equivalent is ``SubtractModel`` below with a model that carries only those terms."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from ..constants import ATOMIC_NUMBER
from ..nn import SelfEnergy, SpeciesConverter

__all__ = ["Transform", "Identity", "identity", "SubtractSAE", "AtomicNumbersToIndices", "Compose", "SubtractModel"]


class Transform(torch.nn.Module):
    """Base class: ``forward(properties) -> properties``.  ``atomic_numbers``: the elements a transform is limited to, in
    its order, or None (transforms.py:43-64)."""

    atomic_numbers: tp.Optional[Tensor]

    def __init__(self, *args: tp.Any, **kwargs: tp.Any) -> None:
        super().__init__()

    def forward(self, properties: tp.Dict[str, Tensor]) -> tp.Dict[str, Tensor]:
        raise NotImplementedError("Must be overriden by subclasses")


class Identity(Transform):
    """Pass-through."""

    def __init__(self) -> None:
        super().__init__()
        self.atomic_numbers = None

    def forward(self, properties: tp.Dict[str, Tensor]) -> tp.Dict[str, Tensor]:
        return properties


identity = Identity()


class SubtractSAE(Transform):
    """energies -= sum of the atoms' self energies (transforms.py:154-168); "species" holds atomic numbers."""

    def __init__(self, symbols: tp.Sequence[str], self_energies: tp.Sequence[float]) -> None:
        super().__init__()
        self._shifter = SelfEnergy(symbols, self_energies)
        self._converter = SpeciesConverter(symbols)
        self.atomic_numbers = torch.tensor([ATOMIC_NUMBER[s] for s in symbols], dtype=torch.long)

    def forward(self, properties: tp.Dict[str, Tensor]) -> tp.Dict[str, Tensor]:
        properties["energies"] -= self._shifter(self._converter(properties["species"]))
        return properties


class AtomicNumbersToIndices(Transform):
    """species: atomic numbers -> element indices of ``symbols`` (transforms.py:171-191); normally the LAST transform."""

    def __init__(self, symbols: tp.Sequence[str]) -> None:
        super().__init__()
        self.atomic_numbers = torch.tensor([ATOMIC_NUMBER[s] for s in symbols], dtype=torch.long)
        self.converter = SpeciesConverter(symbols)

    def forward(self, properties: tp.Dict[str, Tensor]) -> tp.Dict[str, Tensor]:
        properties["species"] = self.converter(properties["species"])
        return properties


class SubtractModel(Transform):
    """energies (and forces) -= those of ``model`` for the batch: the role of the reference's SubtractEnergyAndForce with an
    engine-backed model, e.g. one that carries only pair potentials.  ``model.energies_and_forces(species, coordinates)`` is
    called with the batch's atomic numbers (``periodic_table_index=True`` models)."""

    def __init__(self, model, subtract_force: bool = True) -> None:
        super().__init__()
        self.model = model
        self.subtract_force = subtract_force
        self.atomic_numbers = getattr(model, "atomic_numbers", None)

    def forward(self, properties: tp.Dict[str, Tensor]) -> tp.Dict[str, Tensor]:
        out = self.model.energies_and_forces(properties["species"], properties["coordinates"])
        properties["energies"] -= out.energies.to(properties["energies"].dtype)
        if self.subtract_force:
            properties["forces"] -= out.forces.to(properties["forces"].dtype)
        return properties


class Compose(Transform):
    """Chain transforms in order (transforms.py:194-230); all limited ones must agree on their elements."""

    def __init__(self, transforms: tp.Sequence[Transform]) -> None:
        super().__init__()
        limited = [t.atomic_numbers for t in transforms if t.atomic_numbers is not None]
        if limited and not all(a.shape == limited[0].shape and bool((a.cpu() == limited[0].cpu()).all()) for a in limited):
            raise ValueError("All composed transforms must support the same atomic numbers")
        self.atomic_numbers = limited[0] if limited else None
        self.transforms = torch.nn.ModuleList(transforms)

    def forward(self, properties: tp.Dict[str, Tensor]) -> tp.Dict[str, Tensor]:
        for t in self.transforms:
            properties = t(properties)
        return properties

    def __repr__(self) -> str:
        return "".join([type(self).__name__, "("] + [f"\n    {t}" for t in self.transforms] + ["\n)"])
