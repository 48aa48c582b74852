"""ASE calculator interface (the reference's ``torchani.ase.Calculator``, torchani/ase.py:32-173, returned by
``model.ase()``): energies in eV, forces in eV/Angstrom, stress in eV/Angstrom^3 from the fused HIP path.

ASE itself is an optional dependency (absent in the build image): with ASE installed the class derives from
``ase.calculators.calculator.Calculator``; without it the same ``calculate(atoms, properties)`` / ``results`` protocol
works on any object with ``get_atomic_numbers / get_positions / get_cell / get_pbc / get_volume`` (how the tests
drive it).
"""
from __future__ import annotations

import typing as tp

import numpy as np
import torch

HARTREE_TO_EV = 27.211386024367243   # (= ase.units.Hartree, what the reference's calculator converts with: ase.py:24,134-168)

try:  # pragma: no cover - ASE is not installed in the build image
    import ase.units
    from ase.calculators.calculator import Calculator as _Base
    from ase.calculators.calculator import all_changes

    HARTREE_TO_EV = ase.units.Hartree
    ASE_IS_AVAILABLE = True
except ImportError:
    ASE_IS_AVAILABLE = False
    all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]

    class _Base:   # the part of ase.calculators.calculator.Calculator this module relies on
        def __init__(self) -> None:
            self.results: tp.Dict[str, tp.Any] = {}
            self.atoms = None

        def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes) -> None:
            if atoms is not None:
                self.atoms = atoms


class Calculator(_Base):
    """``stress_kind``: only "fdotr" is native here (ase.py:164-168; for a periodic cell it equals the reference's
    default "scaling" stress, see tests/golden/gen_golden_stress.py); "scaling" is accepted as an alias."""

    implemented_properties = ["energy", "free_energy", "forces", "stress"]

    def __init__(self, model, overwrite: bool = False, stress_kind: str = "fdotr") -> None:
        super().__init__()
        if not model.periodic_table_index:
            raise ValueError("ASE models must have periodic_table_index=True")   # ase.py:69-70
        if stress_kind not in ("fdotr", "scaling"):
            raise ValueError(f"Unsupported stress kind {stress_kind}")
        self.model = model
        self.device = next(model.parameters()).device
        self.overwrite = overwrite
        self.stress_kind = stress_kind

    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes) -> None:
        super().calculate(atoms, properties, system_changes)
        assert self.atoms is not None
        # the SAME species tensor for as long as the atomic numbers stay the same: the model keeps per-system decisions
        # (species numbering, tile hints, atom-order probe, HIP graphs of small systems) tied to that tensor
        numbers = np.asarray(self.atoms.get_atomic_numbers())
        held = self.__dict__.get("_species_held")
        if held is None or held[0].shape != numbers.shape or not np.array_equal(held[0], numbers):
            held = (numbers.copy(), torch.as_tensor(numbers, dtype=torch.long, device=self.device).unsqueeze(0))
            self.__dict__["_species_held"] = held
        species = held[1]
        coords = torch.as_tensor(np.asarray(self.atoms.get_positions()), dtype=torch.float32, device=self.device)
        cell_obj = self.atoms.get_cell(complete=True)
        cell = torch.as_tensor(np.asarray(getattr(cell_obj, "array", cell_obj)), dtype=torch.float32,
                               device=self.device)
        pbc = tuple(bool(b) for b in self.atoms.get_pbc())
        if any(pbc) and self.overwrite:
            # wrap into the central cell and write the wrapped positions back (ase.py:101-104)
            frac = torch.linalg.solve(cell.double().T, coords.double().T).T
            mask = torch.tensor(pbc, device=self.device)
            frac = torch.where(mask, frac - torch.floor(frac), frac)
            coords = (frac @ cell.double()).float()
            self.atoms.set_positions(coords.cpu().numpy())
        use_cell = any(pbc)
        needs_stress = "stress" in properties
        out = self.model.energies_and_forces(species, coords.unsqueeze(0), cell if use_cell else None,
                                             pbc if use_cell else None, stress=needs_stress)
        energy = float(out.energies.item()) * HARTREE_TO_EV
        self.results["energy"] = energy
        self.results["free_energy"] = energy
        if "forces" in properties:
            self.results["forces"] = out.forces[0].double().cpu().numpy() * HARTREE_TO_EV
        if needs_stress:
            volume = float(self.atoms.get_volume())
            self.results["stress"] = out.virial.cpu().numpy() * (HARTREE_TO_EV / volume)
