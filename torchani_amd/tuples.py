"""Named tuples returned by the public API (mirrors torchani/tuples.py:32-36,92-96)."""
from __future__ import annotations

import typing as tp

from torch import Tensor


class Neighbors(tp.NamedTuple):
    """Half neighbor list of the reference (neighbors.py:22-29): indices [2, P] into the flattened atoms,
    distances [P], diff_vectors [P, 3] = r[indices[0]] - r[indices[1]] (+ image shift)."""

    indices: Tensor
    distances: Tensor
    diff_vectors: Tensor


class SpeciesAEV(tp.NamedTuple):
    species: Tensor
    aevs: Tensor


class SpeciesEnergies(tp.NamedTuple):
    species: Tensor
    energies: Tensor


class SpeciesEnergiesAtomicCharges(tp.NamedTuple):
    """Output of ANIq models (torchani/tuples.py:59-62)."""

    species: Tensor
    energies: Tensor
    atomic_charges: Tensor


class EnergiesScalars(tp.NamedTuple):
    """Return type of ANI.compute_from_neighbors / compute_from_external_neighbors (torchani/tuples.py:8-10)."""

    energies: Tensor
    scalars: tp.Optional[Tensor] = None


class EnergiesForces(tp.NamedTuple):
    """What grad.energies_and_forces returns: the reference's two-field tuple (torchani/tuples.py:13-15)."""

    energies: Tensor
    forces: Tensor


class FusedEnergiesForces(tp.NamedTuple):
    """Result of the fused engine path (ANI.energies_and_forces): energies [C] float64 Hartree, forces [C,A,3] float32 Ha/A,
    atomic_energies [C,A] float32 (network part only, no self energies)."""

    energies: Tensor
    forces: Tensor
    atomic_energies: Tensor
    virial: tp.Optional[Tensor] = None   # [3,3] float64 Hartree (stress=True): dE/d strain; stress = virial / volume


class SpeciesForces(tp.NamedTuple):
    """members_forces: energies [M, C], forces [M, C, A, 3] (torchani/tuples.py)."""

    species: Tensor
    energies: Tensor
    forces: Tensor


class SpeciesEnergiesQBC(tp.NamedTuple):
    species: Tensor
    energies: Tensor
    qbcs: Tensor


class AtomicStdev(tp.NamedTuple):
    species: Tensor
    energies: Tensor
    stdev_atomic_energies: Tensor


class ForceMagnitudes(tp.NamedTuple):
    species: Tensor
    magnitudes: Tensor


class ForceStdev(tp.NamedTuple):
    species: Tensor
    magnitudes: Tensor
    relative_stdev: Tensor
    relative_range: Tensor
