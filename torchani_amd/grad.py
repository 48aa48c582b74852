"""Autograd helpers with the reference's names (torchani/grad.py:42-64,263-290)."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor


def forces(energies: Tensor, coords: Tensor, retain_graph: tp.Optional[bool] = None) -> Tensor:
    """forces = -d(sum energies)/d coords (grad.py:57-64)."""
    (g,) = torch.autograd.grad(energies.sum(), coords, retain_graph=retain_graph)
    return -g


def energies_and_forces(model, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None,
                        pbc: tp.Optional[Tensor] = None) -> tp.Tuple[Tensor, Tensor]:
    """(energies, forces) through torch.autograd, restoring coords.requires_grad (grad.py:263-290)."""
    saved = coords.requires_grad
    coords.requires_grad_(True)
    energies = model((species, coords), cell, pbc).energies
    f = forces(energies, coords)
    coords.requires_grad_(saved)
    return energies.detach(), f
