"""Autograd helpers with the reference's names and behaviour (torchani/grad.py:42-86,263-399).

Hessians (grad.py:86-150,239-260) need second derivatives with respect to the coordinates, which the HIP engine does
not provide (its second-order pass serves training on forces: parameters only): those entry points raise;
``numerical_hessians`` differentiates the analytic forces numerically instead, ``vibrational_analysis`` is the reference's.
"""
from __future__ import annotations

import math
import typing as tp

import torch
from torch import Tensor


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


calc_forces = forces
calc_grads = grads


def forces_for_training(energies: Tensor, coordinates: Tensor) -> Tensor:
    """Forces that can be differentiated again with respect to the parameters (grad.py:82-83): the engine's
    double-backward Functions answer with anihip_aev_jvp / anihip_mlp_tangent_weight_grads."""
    return forces(energies, coordinates, retain_graph=True, create_graph=True)


def _no_hessians(*args, **kwargs):
    raise NotImplementedError("hessians need second derivatives with respect to the coordinates, which the HIP engine "
                              "does not provide")


forces_and_hessians = hessians = energies_forces_and_hessians = _no_hessians   # (analytic: see numerical_hessians below)


def numerical_hessians(model, species: Tensor, coordinates: Tensor, cell: tp.Optional[Tensor] = None,
                       pbc: tp.Optional[tp.Sequence[bool]] = None, step: float = 0.01) -> Tensor:
    """Hessians [C, 3A, 3A] (float64, Hartree / A^2) by central differences of the analytic forces: for every molecule the
    6 A displaced copies (each coordinate by +-``step`` A) are ONE batch for ``model.energies_and_forces``, which is what
    the engine is good at.  The reference differentiates twice through autograd (grad.py:86-149); the HIP engine has no
    second derivatives with respect to the coordinates, and an fp32 force field differentiated numerically is what e.g.
    ASE's vibrations module does with any calculator.  Expect ~1e-4 Ha / A^2 of noise (fp32 forces / 2 step): good for
    stretches and bends, not for the softest modes.  Rows / columns of padding atoms are zero.  The result is
    symmetrized.  ``model`` needs ``energies_and_forces(species, coordinates, cell, pbc)`` returning ``.forces``."""
    if species.dim() != 2 or coordinates.shape != (species.shape[0], species.shape[1], 3):
        raise ValueError("expected species [C, A] and coordinates [C, A, 3]")
    if not step > 0:
        raise ValueError("step must be positive")
    C, A = species.shape
    n = 3 * A
    out = torch.zeros((C, n, n), dtype=torch.float64, device=coordinates.device)
    eye = torch.eye(n, dtype=coordinates.dtype, device=coordinates.device).view(n, A, 3) * step
    for c in range(C):
        real = (species[c] >= 0).repeat_interleave(3)                         # [3A]
        x = coordinates[c].detach().unsqueeze(0)
        disp = torch.cat([x + eye, x - eye], dim=0)                           # [2 * 3A, A, 3]: +k ..., -k ...
        f = model.energies_and_forces(species[c].unsqueeze(0).expand(2 * n, A).contiguous(), disp.contiguous(), cell,
                                      pbc).forces.to(torch.float64).reshape(2 * n, n)
        h = -(f[:n] - f[n:]) / (2.0 * step)                                   # h[k, l] = d2E / dx_k dx_l
        h = 0.5 * (h + h.t())
        out[c] = h * (real.unsqueeze(0) & real.unsqueeze(1))
    return out


def vibrational_analysis(masses: Tensor, hessian: Tensor, mode_kind: str = "mdu", unit: str = "cm^-1"):
    """Vibrational wavenumbers, normal modes, force constants (mDyne / A) and reduced masses (amu) of ONE molecule from its
    Hessian [1, 3A, 3A] (Hartree / A^2) and masses [1, A] (amu) -- grad.py:152-236.  The mass-scaled Hessian
    T^-1/2 H T^-1/2 is diagonalized; ``mode_kind``: "mwn" (mass weighted, orthonormal), "mdu" (mass deweighted,
    unnormalized: ASE), "mdn" (mass deweighted, normalized: Gaussian, ORCA).  Imaginary frequencies come out negative;
    the six smallest belong to translations and rotations.  ``unit``: "cm^-1" or "meV"."""
    from .tuples import VibAnalysis
    from .units import mhessian2fconst, sqrt_mhessian2invcm, sqrt_mhessian2milliev

    if unit == "meV":
        convert = sqrt_mhessian2milliev
    elif unit == "cm^-1":
        convert = sqrt_mhessian2invcm
    else:
        raise ValueError("Only meV and cm^-1 are supported right now")
    if hessian.dim() != 3 or hessian.shape[0] != 1:
        raise ValueError("The input should contain only one molecule")
    inv_sqrt_m = (1 / masses.sqrt()).repeat_interleave(3, dim=1)              # [1, 3A]
    scaled = (hessian * inv_sqrt_m.unsqueeze(1) * inv_sqrt_m.unsqueeze(2)).squeeze(0)
    eigenvalues, eigenvectors = torch.linalg.eigh(scaled)
    mw_normalized = eigenvectors.t()                                          # (the modes are the COLUMNS of eigenvectors)
    md_unnormalized = mw_normalized * inv_sqrt_m
    norm_factors = 1 / torch.linalg.norm(md_unnormalized, dim=1)              # sqrt(amu)
    rmasses = norm_factors ** 2
    fconstants = mhessian2fconst(eigenvalues) * rmasses
    kind = mode_kind.lower()
    if kind in ("mdn", "mass-deweighted-normalized"):
        modes = md_unnormalized * norm_factors.unsqueeze(1)
    elif kind in ("mdu", "mass-deweighted-unnormalized"):
        modes = md_unnormalized
    elif kind in ("mwn", "mass-weighted-normalized"):
        modes = mw_normalized
    else:
        raise ValueError(f"Incorrect mode kind {mode_kind}")
    freqs = convert(eigenvalues.abs().sqrt() / (2 * math.pi) * torch.sign(eigenvalues))
    return VibAnalysis(freqs, modes.reshape(eigenvalues.numel(), -1, 3), fconstants, rmasses)


def energies_and_forces(model, species: Tensor, coordinates: Tensor, cell: tp.Optional[Tensor] = None,
                        pbc: tp.Optional[Tensor] = None, charge: int = 0, atomic: bool = False,
                        ensemble_values: bool = False, keep_vars: bool = False) -> tp.Tuple[Tensor, Tensor]:
    """(energies, forces) through torch.autograd, restoring coordinates.requires_grad (grad.py:263-290)."""
    saved = coordinates.requires_grad
    coordinates.requires_grad_(True)
    energies = model((species, coordinates), cell, pbc, atomic=atomic, ensemble_values=ensemble_values).energies
    f = forces(energies, coordinates)
    coordinates.requires_grad_(saved)
    if not keep_vars:
        energies = energies.detach()
    return energies, f


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
