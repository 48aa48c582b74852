"""Cutoff (envelope) functions as callable modules under the reference's names (torchani/cutoffs.py:17-143).

The HIP kernels evaluate the cosine and the smooth (order 2) envelope themselves; these modules are the host-side
counterparts -- ``CutoffSmooth()(distances, cutoff)`` on any tensor -- and are accepted wherever the package takes a
``cutoff_fn`` (``kernel_name`` maps an object to the name the kernels know, and refuses the ones they do not cover)."""
from __future__ import annotations

import math
import typing as tp

import torch
from torch import Tensor

__all__ = ["Cutoff", "CutoffDummy", "CutoffBiweight", "CutoffTriweight", "CutoffCosine", "CutoffSmooth", "parse_cutoff_fn",
           "kernel_name"]


class Cutoff(torch.nn.Module):
    """Base class: ``forward(distances, cutoff)`` -> factors in [0, 1] that take a pair term smoothly to zero at the cutoff."""

    _kernel_name: str = ""

    def __init__(self, *args: tp.Any, **kwargs: tp.Any) -> None:
        super().__init__()
        self._fn_params = args + tuple(kwargs.values())

    def is_same(self, other: object) -> bool:
        return isinstance(other, Cutoff) and type(self) is type(other) and self._fn_params == other._fn_params

    def forward(self, distances: Tensor, cutoff: float) -> Tensor:
        raise NotImplementedError


class CutoffDummy(Cutoff):
    """Ones."""

    _kernel_name = "dummy"

    def forward(self, distances: Tensor, cutoff: float) -> Tensor:
        return torch.ones_like(distances)


class CutoffBiweight(Cutoff):
    """(1 - (r / rc)^2)^2"""

    def forward(self, distances: Tensor, cutoff: float) -> Tensor:
        return (1 - (distances / cutoff) ** 2) ** 2


class CutoffTriweight(Cutoff):
    """(1 - (r / rc)^2)^3"""

    def forward(self, distances: Tensor, cutoff: float) -> Tensor:
        return (1 - (distances / cutoff) ** 2) ** 3


class CutoffCosine(Cutoff):
    """0.5 cos(pi r / rc) + 0.5 -- ANI-1x / ANI-2x (cutoffs.py:70-81)"""

    _kernel_name = "cosine"

    def forward(self, distances: Tensor, cutoff: float) -> Tensor:
        return 0.5 * torch.cos(distances * (math.pi / cutoff)) + 0.5


class CutoffSmooth(Cutoff):
    """exp(1 - 1 / max(eps, 1 - (r / rc)^n)), infinitely differentiable -- the newer models (cutoffs.py:84-107).  The
    kernels implement order 2 with the default eps."""

    def __init__(self, order: int = 2, eps: float = 1.0e-10) -> None:
        super().__init__(order, eps)
        self.order, self.eps = order, eps
        self._kernel_name = "smooth" if (order == 2 and eps == 1.0e-10) else ""

    def forward(self, distances: Tensor, cutoff: float) -> Tensor:
        return torch.exp(1 - 1 / (1 - (distances / cutoff) ** self.order).clamp(min=self.eps))

    def extra_repr(self) -> str:
        return f"order={self.order}, eps={self.eps:.1e}"


_BY_NAME = {"dummy": CutoffDummy, "cosine": CutoffCosine, "smooth": CutoffSmooth, "biweight": CutoffBiweight,
            "triweight": CutoffTriweight}


def parse_cutoff_fn(cutoff_fn: tp.Union[str, Cutoff], global_cutoff: tp.Optional[Cutoff] = None) -> Cutoff:
    """Name or object -> Cutoff object (cutoffs.py:124-143); "global" stands for ``global_cutoff``."""
    if isinstance(cutoff_fn, str) and cutoff_fn == "global":
        assert global_cutoff is not None
        cutoff_fn = global_cutoff
    if isinstance(cutoff_fn, str) and cutoff_fn in _BY_NAME:
        return _BY_NAME[cutoff_fn]()
    if not isinstance(cutoff_fn, Cutoff):
        raise ValueError(f"Unsupported cutoff fn: {cutoff_fn}")
    return cutoff_fn


def kernel_name(cutoff_fn: tp.Union[str, Cutoff]) -> str:
    """The name under which the HIP kernels know ``cutoff_fn`` ("cosine", "smooth", "dummy"), from a name or a Cutoff
    object; ValueError for envelopes the kernels do not implement."""
    if isinstance(cutoff_fn, Cutoff):
        if not cutoff_fn._kernel_name:
            raise ValueError(f"the HIP kernels do not implement {cutoff_fn!r}: 'cosine' and 'smooth' (order 2) envelopes only")
        return cutoff_fn._kernel_name
    return str(cutoff_fn)
