"""Dispatcher registration of the C ABI: ``torch.ops.anihip.{nbr_rows, aev, aev_backward, mlp}``.

The reference's boundary is three sets of dispatcher operators + a custom class (``TORCH_LIBRARY(cuaev, ...)``,
csrc/cuaev.cpp:246-294; ``cell_list``, csrc/cell_list.cpp:351-363; ``mnp``, csrc/mnp.cpp:273-280), which is what
makes its native path scriptable and traceable (tests/test_cuaev.py:104-142, tests/test_pt2.py:44-60).  This is
the same thing for libanihip: thin ``torch.library`` operators that forward to the C ABI through
torchani_amd.engine, with fake (meta) kernels and autograd formulas, so ``torch.jit.script`` sees
``anihip::aev`` nodes and ``torch.compile(fullgraph=True)`` keeps the whole energy graph.

The role of ``torch.classes.cuaev.CuaevComputer`` (the bag of AEV constants handed to every call) is played by an
integer handle from ``register_engine`` / ``register_networks``.
"""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor

from . import _lib
from .engine import AevEngine, NeighborRows, PackedNetworks

_engines: tp.List[AevEngine] = []
_networks: tp.List[PackedNetworks] = []
_MODES = ("auto", "batch", "cell")


def register_engine(eng: AevEngine) -> int:
    """Handle of an AevEngine (constants of one AEVComputer) for the operators below."""
    if eng not in _engines:
        _engines.append(eng)
    return _engines.index(eng)


def register_networks(packed: PackedNetworks) -> int:
    if packed not in _networks:
        _networks.append(packed)
    return _networks.index(packed)


def _pbc(pbc_mask: int):
    return tuple(bool(pbc_mask & (1 << k)) for k in range(3)) if pbc_mask else None


# ---- neighbor rows: cell_list::cell_list / cuaev's internal lists ------------------------------------------------
@torch.library.custom_op("anihip::nbr_rows", mutates_args=())
def nbr_rows(species: Tensor, coords: Tensor, cell: tp.Optional[Tensor], pbc_mask: int, mode: int, row_cap: int,
             engine: int) -> tp.Tuple[Tensor, Tensor, Tensor]:
    """species int32 [C, A], coords fp32 [C, A, 3] -> (meta [N, 6] int32, ent [N * row_cap, 4] fp32, status [8]);
    mode 0 auto, 1 all pairs per molecule, 2 cell list (include/anihip.h: anihip_nbr_build_batch / _cell)."""
    r = _engines[engine].neighbors(species, coords.detach(), cell, _pbc(pbc_mask), mode=_MODES[mode], row_cap=row_cap)
    return r.meta, r.ent, r.status


@nbr_rows.register_fake
def _(species, coords, cell, pbc_mask, mode, row_cap, engine):
    n = species.numel()
    return (species.new_empty((n, _lib.META_WORDS), dtype=torch.int32),
            coords.new_empty((max(n, 1) * row_cap, 4), dtype=torch.float32),
            species.new_empty((_lib.STATUS_WORDS,), dtype=torch.int32))


# ---- AEV forward / backward: cuaev::run + CuaevAutograd (csrc/cuaev.cpp:120-139,189-203) --------------------------
def _rows(meta: Tensor, ent: Tensor, status: Tensor, n: int) -> NeighborRows:
    return NeighborRows(meta, ent, status, ent.shape[0] // max(n, 1), 0, n)


@torch.library.custom_op("anihip::aev_from_rows", mutates_args=())
def aev_from_rows(species: Tensor, coords: Tensor, meta: Tensor, ent: Tensor, status: Tensor, engine: int) -> Tensor:
    """AEVs [C, A, L] from neighbor rows; coords only carries the gradient (displacements live in ent)."""
    eng = _engines[engine]
    C, A = species.shape
    return eng.forward(species, _rows(meta, ent, status, C * A)).view(C, A, eng.L)


@aev_from_rows.register_fake
def _(species, coords, meta, ent, status, engine):
    return coords.new_empty((species.shape[0], species.shape[1], _engines[engine].L), dtype=torch.float32)


@torch.library.custom_op("anihip::aev_backward", mutates_args=())
def aev_backward(grad_aev: Tensor, species: Tensor, meta: Tensor, ent: Tensor, status: Tensor, engine: int) -> Tensor:
    """grad_coords [C, A, 3] = J^T grad_aev (anihip_aev_backward)."""
    C, A = species.shape
    g = grad_aev.detach().to(torch.float32).contiguous()
    return _engines[engine].backward(species, _rows(meta, ent, status, C * A), g).view(C, A, 3)


@aev_backward.register_fake
def _(grad_aev, species, meta, ent, status, engine):
    return grad_aev.new_empty((species.shape[0], species.shape[1], 3), dtype=torch.float32)


def _aev_setup(ctx, inputs, output):
    species, _, meta, ent, status, engine = inputs
    ctx.save_for_backward(species, meta, ent, status)
    ctx.engine = engine


def _aev_bwd(ctx, grad_aev):
    species, meta, ent, status = ctx.saved_tensors
    return None, torch.ops.anihip.aev_backward(grad_aev, species, meta, ent, status, ctx.engine), None, None, None, None


aev_from_rows.register_autograd(_aev_bwd, setup_context=_aev_setup)


def aev(species: Tensor, coords: Tensor, cell: tp.Optional[Tensor], pbc_mask: int, mode: int, row_cap: int,
        engine: int) -> Tensor:
    """``cuaev::run`` in one call: neighbor rows + AEVs, differentiable with respect to coords."""
    # (the rows carry no gradient: like cuaev's own lists, displacements are not differentiated -- the analytic
    # backward of aev_from_rows is the whole derivative with respect to coords)
    meta, ent, status = torch.ops.anihip.nbr_rows(species, coords.detach(), cell, pbc_mask, mode, row_cap, engine)
    return torch.ops.anihip.aev_from_rows(species, coords, meta, ent, status, engine)


# ---- ensemble of atomic networks: mnp::run (csrc/mnp.cpp:238-265) -------------------------------------------------
@torch.library.custom_op("anihip::mlp", mutates_args=())
def mlp(species: Tensor, aevs: Tensor, networks: int) -> tp.Tuple[Tensor, Tensor]:
    """(atomic energies [C, A] = ensemble mean, d atomic_e / d aev [C, A, L]) in one pass, like MultiNetFunction's
    forward which keeps the input gradient for its backward (csrc/mnp.cpp:32-136)."""
    C, A = species.shape
    a = aevs.detach().to(torch.float32).contiguous().view(C * A, -1)
    e, g, _ = _networks[networks].forward_backward(species, a, want_grad=True)
    return e.view(C, A), g.view(C, A, -1)


@mlp.register_fake
def _(species, aevs, networks):
    return aevs.new_empty(species.shape, dtype=torch.float32), torch.empty_like(aevs, dtype=torch.float32)


def _mlp_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _mlp_bwd(ctx, grad_e, grad_unused):
    (daev,) = ctx.saved_tensors
    return None, grad_e.unsqueeze(-1) * daev, None


mlp.register_autograd(_mlp_bwd, setup_context=_mlp_setup)


class CompiledEnergy(torch.nn.Module):
    """species (element indices), coords -> molecular energies through the registered operators only: the module that
    ``torch.jit.script`` / ``torch.compile(fullgraph=True)`` can take whole (the reference checks the same for its
    native path, tests/test_pt2.py:44-60)."""

    def __init__(self, model) -> None:
        super().__init__()
        aevc = model.aev_computer
        self.engine = register_engine(aevc.engine())
        self.networks = -1
        self._model = [model]   # not a submodule: parameters stay with the caller's model
        self.mode = _MODES.index(aevc.neighbor_mode)
        self.row_cap = int(aevc.row_capacity)
        self.register_buffer("sae", model.energy_shifter.self_energies.detach().to(torch.float64).clone())

    def bind(self, device: torch.device) -> "CompiledEnergy":
        self.networks = register_networks(self._model[0].neural_networks._pack(device))
        return self

    def forward(self, species: Tensor, coords: Tensor) -> Tensor:
        sp32 = species.to(torch.int32)
        a = aev(sp32, coords, None, 0, self.mode, self.row_cap, self.engine)
        e_atom, _ = torch.ops.anihip.mlp(sp32, a, self.networks)
        shift = self.sae[species.clamp(min=0)] * (species >= 0)
        return (e_atom.to(torch.float64) * (species >= 0) + shift).sum(dim=1)
