"""Self atomic energies from a dataset of batches (torchani/sae_estimation.py:20-131): the per-element energies that best
explain the molecular energies linearly, which a training run subtracts first (transforms.SubtractSAE) and a model adds back
(SelfEnergy).  ``dataset`` is any iterable of property dictionaries with "species" (atomic numbers, padding -1) and
"energies"; if it has a ``transform`` attribute (the reference's BatchedDataset protocol) the batches are expected to come out
transformed by it, and it is set to AtomicNumbersToIndices for the pass and restored.  Pure tensor code."""
from __future__ import annotations

import math
import typing as tp

import torch
from torch import Tensor

from .transforms import AtomicNumbersToIndices

__all__ = ["exact_saes", "approx_saes"]


def _batches(dataset, symbols: tp.Sequence[str], fraction: float, device) -> tp.Iterator[tp.Tuple[Tensor, Tensor]]:
    """(element counts [c, S] float32, energies [c] float32) of the first ceil(len * fraction) batches."""
    to_idx = AtomicNumbersToIndices(symbols)
    has_transform = hasattr(dataset, "transform")
    old = dataset.transform if has_transform else None
    if has_transform:
        dataset.transform = to_idx
    try:
        n_use = math.ceil(len(dataset) * fraction)
        for j, properties in enumerate(dataset):
            if j >= n_use:
                break
            if not has_transform:
                properties = to_idx(dict(properties))
            species = properties["species"].to(device)
            counts = torch.stack([(species == k).sum(-1) for k in range(len(symbols))], dim=1).float()
            yield counts, properties["energies"].to(dtype=torch.float, device=device)
    finally:
        if has_transform:
            dataset.transform = old


def exact_saes(dataset, symbols: tp.Sequence[str], fraction: float = 1.0, fit_intercept: bool = False,
               device=None) -> tp.Tuple[Tensor, tp.Optional[Tensor]]:
    """Least-squares self energies [S] (and the intercept, or None) -- sae_estimation.py:20-75.  (With ``fit_intercept`` the
    design matrix gets a column of ones; the reference appends a ROW there, which its solver rejects.)"""
    counts, energies = zip(*_batches(dataset, symbols, fraction, device))
    a, b = torch.cat(counts, dim=0), torch.cat(energies, dim=0)
    if fit_intercept:
        a = torch.cat([a, torch.ones((a.shape[0], 1), dtype=a.dtype, device=a.device)], dim=1)
    x = torch.linalg.lstsq(a, b.unsqueeze(-1), driver="gels").solution.squeeze(-1)
    if fit_intercept:
        return x[:len(symbols)], x[len(symbols)]
    return x, None


def approx_saes(dataset, symbols: tp.Sequence[str], fraction: float = 1.0, fit_intercept: bool = False, device=None,
                max_epochs: int = 1, lr: float = 0.01) -> tp.Tuple[Tensor, tp.Optional[Tensor]]:
    """The same by stochastic gradient descent over the batches (sae_estimation.py:78-131), for datasets too large for one
    least-squares solve."""
    m = torch.nn.Parameter(torch.ones(len(symbols), dtype=torch.float, device=device))
    b = torch.nn.Parameter(torch.zeros(1, dtype=torch.float, device=device)) if fit_intercept else None
    opt = torch.optim.SGD([m] + ([b] if b is not None else []), lr=lr)
    for _ in range(max_epochs):
        for counts, energies in _batches(dataset, symbols, fraction, device):
            pred = (counts * m + (b if b is not None else 0.0)).sum(-1)   # (the reference's _LinearModel, intercept per column)
            loss = (energies - pred).pow(2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
    return m.detach().cpu(), (b.detach().cpu() if b is not None else None)
