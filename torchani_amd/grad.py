"""Autograd helpers with the reference's names and behaviour (torchani/grad.py:42-86,263-399).

Hessians (grad.py:86-150,239-260) need second derivatives with respect to the coordinates, which the HIP engine does
not provide (its second-order pass serves training on forces: parameters only): ``hessians`` / ``forces_and_hessians``
raise and say so.  NOT here (removed in round 3 as outside the hot path, SURVEY section 2 "OUT OF SCOPE"): the reference's
``vibrational_analysis`` / ``VibAnalysis`` (grad.py:153-236), a numerical Hessian, and the modules ``torchani.units``,
``torchani.cutoffs`` (the cutoff envelopes live in the AEV kernels: ``AEVComputer(..., cutoff_fn="cosine" | "smooth")``,
``constants.cutoff_kernel_name``) and ``torchani.sae`` (``nn.SelfEnergy`` is the energy shifter of the models).
"""
from __future__ import annotations

import math
import typing as tp

import torch
from torch import Tensor

from .tuples import EnergiesForces


def forces(energies: Tensor, coordinates: Tensor, retain_graph: tp.Optional[bool] = None,
           create_graph: bool = False) -> Tensor:
    """forces = -d(sum energies)/d coordinates (grad.py:42-64)."""
    if not coordinates.requires_grad:
        raise ValueError("'coordinates' passed to `torchani.grad.forces` must require grad")
    if not coordinates.is_leaf:
        raise ValueError("'coordinates' passed to `torchani.grad` functions must be a 'leaf' Tensor"
                         "(i.e. must not have been modified prior to being used as an input).")
    (g,) = torch.autograd.grad([energies.sum()], [coordinates], retain_graph=retain_graph, create_graph=create_graph)
    return -g


def grads(scalars: Tensor, coords: Tensor, retain_graph: tp.Optional[bool] = None,
          create_graph: bool = False) -> Tensor:
    """Alias of forces with the sign flipped (grad.py:68-74)."""
    return -forces(scalars, coords, retain_graph, create_graph)


__all__ = ["single_point", "forces_for_training", "energies_and_forces", "forces", "grads", "calc_forces", "calc_grads",
           "hessians", "forces_and_hessians", "energies_forces_and_hessians"]   # (the last three raise: see the module docstring)

calc_forces = forces
calc_grads = grads


def forces_for_training(energies: Tensor, coordinates: Tensor) -> Tensor:
    """Forces that can be differentiated again with respect to the parameters (grad.py:82-83): the engine's
    double-backward Functions answer with anihip_aev_jvp / anihip_mlp_tangent_weight_grads."""
    return forces(energies, coordinates, retain_graph=True, create_graph=True)


def _no_hessians(*args, **kwargs):
    raise NotImplementedError("hessians need second derivatives with respect to the coordinates, which the HIP engine "
                              "does not provide")


forces_and_hessians = hessians = energies_forces_and_hessians = _no_hessians


def energies_and_forces(model, species: Tensor, coordinates: Tensor, cell: tp.Optional[Tensor] = None,
                        pbc: tp.Optional[Tensor] = None, retain_graph: tp.Optional[bool] = None,
                        create_graph: bool = False, charge: int = 0, atomic: bool = False,
                        ensemble_values: bool = False, keep_vars: bool = True) -> EnergiesForces:
    """``EnergiesForces(energies, forces)`` through torch.autograd, restoring coordinates.requires_grad: the signature, the
    leaf check and the result of the reference (grad.py:263-290) -- ``retain_graph`` / ``create_graph`` go to the force
    derivative (``create_graph=True``: forces that can be trained on, tools/training-aev-benchmark.py:139-140), the energies
    keep their graph like the reference's.  A standalone pair potential (``torchani_amd.potentials``) is called as
    ``model(species, coordinates, cell, pbc)``.  The keywords behind ``create_graph`` are extensions of this package:
    ``atomic`` / ``ensemble_values`` are handed to the model, ``keep_vars=False`` detaches the energies."""
    from .potentials import _Standalone

    saved = coordinates.requires_grad
    coordinates.requires_grad_(True)
    if not coordinates.is_leaf:
        raise ValueError("'coordinates' passed to `torchani.grad` functions must be a 'leaf' Tensor"
                         "(i.e. must not have been modified prior to being used as an input).")
    if isinstance(model, _Standalone):
        energies = model(species, coordinates, cell, pbc)
    elif atomic or ensemble_values:
        energies = model((species, coordinates), cell, pbc, atomic=atomic, ensemble_values=ensemble_values).energies
    else:
        energies = model((species, coordinates), cell, pbc).energies
    f = forces(energies, coordinates, retain_graph=retain_graph, create_graph=create_graph)
    coordinates.requires_grad_(saved)
    if not keep_vars:
        energies = energies.detach()
    return EnergiesForces(energies, f)


def single_point(model, species: Tensor, coordinates: Tensor, cell: tp.Optional[Tensor] = None,
                 pbc: tp.Optional[Tensor] = None, charge: int = 0, forces: bool = False, hessians: bool = False,
                 atomic_energies: bool = False, atomic_charges: bool = False, atomic_charges_grad: bool = False,
                 ensemble_values: bool = False, keep_vars: bool = False) -> tp.Dict[str, Tensor]:
    """Properties of a batch of molecules as a dictionary (grad.py:293-399): energies, optional forces, atomic
    energies, and -- with ensemble_values -- the member values, their standard deviation and the QBC factors."""
    if hessians:
        _no_hessians()
    if atomic_charges_grad:
        raise NotImplementedError("atomic charges carry no gradient here (the charge networks run the inference kernels)")
    if forces and ensemble_values:
        raise NotImplementedError("forces of ensemble_values=True are not differentiable in the HIP engine")
    saved = coordinates.requires_grad
    if forces:
        coordinates.requires_grad_(True)
    result = model((species, coordinates), cell, pbc, atomic=atomic_energies, ensemble_values=ensemble_values)
    energies = result.energies
    out: tp.Dict[str, Tensor] = {}
    if atomic_charges:   # (ANIq models: models.ANImbis, models.simple_aniq)
        if not hasattr(result, "atomic_charges"):
            coordinates.requires_grad_(saved)
            raise ValueError("Model doesn't support atomic charges")
        out["atomic_charges"] = result.atomic_charges
    if ensemble_values:
        if atomic_energies:
            out["atomic_energies"] = energies.mean(dim=0)
            values = energies.sum(dim=-1)
        else:
            values = energies
        out["energies"] = values.mean(dim=0)
        single = values.shape[0] == 1
        out["ensemble_std"] = values.new_zeros(energies.shape) if single else values.std(dim=0, unbiased=True)
        out["ensemble_values"] = values
        qbc = values.new_zeros(values.shape).squeeze(0) if single else values.std(0, unbiased=True)
        out["qbcs"] = qbc / (species >= 0).sum(dim=1, dtype=energies.dtype).sqrt()
    elif atomic_energies:
        out["energies"] = energies.sum(dim=-1)
        out["atomic_energies"] = energies
    else:
        out["energies"] = energies
    if forces:
        out["forces"] = calc_forces(out["energies"], coordinates)
    coordinates.requires_grad_(saved)
    if not keep_vars:
        out = {k: v.detach() for k, v in out.items()}
    return out
