"""AEVComputer with the reference's call signature, backed by the HIP engine.

Mirrors torchani/aev/_computer.py:73-129 (constructor), :193-249 (forward), :498-666 (like_1x / like_2x /
from_constants) and the buffers of torchani/aev/_terms.py:153-156,288-292 so that a reference state dict
loads unchanged.  ``strategy`` is always the native HIP path: there is no pyaev fallback here.
"""
from __future__ import annotations

import math
import typing as tp
import warnings

import torch
from torch import Tensor

from .constants import AEVConstants, aev_constants_1x, aev_constants_2x, cutoff_kernel_name
from ._lib import MAX_RAD as _MAX_RAD
from .engine import AevEngine, NeighborRows, VerletRows
from .tuples import Neighbors, SpeciesAEV


def _envelope(name: str, distances: Tensor, cutoff: float) -> Tensor:
    # host-side value of the two envelopes the kernels implement (cutoffs.py:80-81, :98-101)
    if name == "cosine":
        return 0.5 * torch.cos(distances * (math.pi / cutoff)) + 0.5
    return torch.exp(1 - 1 / (1 - (distances / cutoff) ** 2).clamp(min=1.0e-10))


class _RadialTerms(torch.nn.Module):
    """Holder of the radial hyper-parameters under the reference's buffer names (eta, shifts)."""

    def __init__(self, eta: float, shifts: tp.Sequence[float], cutoff: float) -> None:
        super().__init__()
        self.register_buffer("eta", torch.tensor([eta], dtype=torch.float))
        self.register_buffer("shifts", torch.tensor(list(shifts), dtype=torch.float))
        self.cutoff = float(cutoff)


class _AngularTerms(torch.nn.Module):
    """Holder of the angular hyper-parameters (eta, zeta, shifts, sections)."""

    def __init__(self, eta: float, zeta: float, shifts: tp.Sequence[float], sections: tp.Sequence[float],
                 cutoff: float) -> None:
        super().__init__()
        self.register_buffer("eta", torch.tensor([eta], dtype=torch.float))
        self.register_buffer("zeta", torch.tensor([zeta], dtype=torch.float))
        self.register_buffer("shifts", torch.tensor(list(shifts), dtype=torch.float))
        self.register_buffer("sections", torch.tensor(list(sections), dtype=torch.float))
        self.cutoff = float(cutoff)


class ANIRadial(_RadialTerms):
    """The ANI two-body expansion 0.25 exp(-eta (r - s)^2) fc(r) as an object (aev/_terms.py:131-241): the hyper-parameters
    an ``Assembler`` / ``AEVComputer.from_terms`` builds the engine's table from, and -- ``module(distances)`` -- a host-side
    evaluation of the terms on any tensor, [pairs] -> [pairs, shifts] (the kernels never call it)."""

    def __init__(self, eta: float, shifts: tp.Sequence[float], cutoff: float, cutoff_fn="cosine") -> None:
        super().__init__(eta, shifts, cutoff)
        self.cutoff_fn = cutoff_kernel_name(cutoff_fn)
        self.num_feats = len(self.shifts)

    def compute(self, distances: Tensor) -> Tensor:
        return 0.25 * torch.exp(-self.eta * (distances - self.shifts.view(1, -1)) ** 2)

    def forward(self, distances: Tensor) -> Tensor:
        assert distances.dim() == 1
        return self.compute(distances.view(-1, 1)) * _envelope(self.cutoff_fn, distances, self.cutoff).view(-1, 1)

    @classmethod
    def cover_linearly(cls, start: float = 0.9, cutoff: float = 5.2, eta: float = 19.7, num_shifts: int = 16,
                       cutoff_fn="cosine") -> "ANIRadial":
        """``num_shifts`` shifts from ``start`` up to (excluding) ``cutoff`` (aev/_terms.py:189-207)."""
        from .constants import linspace

        return cls(eta, linspace(start, cutoff, num_shifts), cutoff, cutoff_fn)

    @classmethod
    def like_1x(cls, start: float = 0.9, cutoff: float = 5.2, eta: float = 16.0, num_shifts: int = 16,
                cutoff_fn="cosine") -> "ANIRadial":
        return cls.cover_linearly(start, cutoff, eta, num_shifts, cutoff_fn)

    @classmethod
    def like_2x(cls, start: float = 0.8, cutoff: float = 5.1, eta: float = 19.7, num_shifts: int = 16,
                cutoff_fn="cosine") -> "ANIRadial":
        return cls.cover_linearly(start, cutoff, eta, num_shifts, cutoff_fn)


class ANIAngular(_AngularTerms):
    """The ANI three-body expansion (aev/_terms.py:244-410): 2 ((1 + cos(theta - theta_s)) / 2)^zeta exp(-eta ((r1 + r2) / 2 -
    s)^2) fc(r1) fc(r2), theta = acos(0.95 cos_angle); ``module(tri_distances [2, T], tri_vectors [2, T, 3])`` evaluates it
    on the host, -> [T, shifts * sections] (shift-major, the layout of an AEV's angular block)."""

    def __init__(self, eta: float, zeta: float, shifts: tp.Sequence[float], sections: tp.Sequence[float], cutoff: float,
                 cutoff_fn="cosine") -> None:
        super().__init__(eta, zeta, shifts, sections, cutoff)
        self.cutoff_fn = cutoff_kernel_name(cutoff_fn)
        self.num_feats = len(self.shifts) * len(self.sections)

    def compute_radial(self, distances_ji: Tensor, distances_jk: Tensor) -> Tensor:
        return torch.exp(-self.eta * ((distances_ji + distances_jk) / 2 - self.shifts.view(1, -1)) ** 2)

    def compute_cos_angles(self, cos_angles: Tensor) -> Tensor:
        return 2 * ((1 + torch.cos(torch.acos(0.95 * cos_angles) - self.sections.view(1, -1))) / 2) ** self.zeta

    def forward(self, tri_distances: Tensor, tri_vectors: Tensor) -> Tensor:
        assert tri_distances.dim() == 2 and tri_vectors.shape == (2, tri_distances.shape[1], 3)
        fc = _envelope(self.cutoff_fn, tri_distances, self.cutoff)
        d = tri_distances.view(2, -1, 1)
        cos_angles = (tri_vectors[0] * tri_vectors[1]).sum(-1, keepdim=True) / torch.clamp(d[0] * d[1], min=1e-10)
        terms = self.compute_radial(d[0], d[1]).unsqueeze(2) * self.compute_cos_angles(cos_angles).unsqueeze(1)
        return terms.reshape(-1, self.num_feats) * (fc[0] * fc[1]).view(-1, 1)

    @classmethod
    def cover_linearly(cls, start: float = 0.9, cutoff: float = 3.5, eta: float = 12.5, zeta: float = 14.1,
                       num_shifts: int = 8, num_sections: int = 4, cutoff_fn="cosine") -> "ANIAngular":
        """Shifts like ANIRadial.cover_linearly, ``num_sections`` angles from pi / (2 n) in steps of pi / n
        (aev/_terms.py:346-366)."""
        from .constants import linspace

        a0 = math.pi / num_sections / 2
        return cls(eta, zeta, linspace(start, cutoff, num_shifts), linspace(a0, math.pi + a0, num_sections), cutoff,
                   cutoff_fn)

    @classmethod
    def like_1x(cls, start: float = 0.9, cutoff: float = 3.5, eta: float = 8.0, zeta: float = 32.0, num_shifts: int = 4,
                num_sections: int = 8, cutoff_fn="cosine") -> "ANIAngular":
        return cls.cover_linearly(start, cutoff, eta, zeta, num_shifts, num_sections, cutoff_fn)

    @classmethod
    def like_2x(cls, start: float = 0.8, cutoff: float = 3.5, eta: float = 12.5, zeta: float = 14.1, num_shifts: int = 8,
                num_sections: int = 4, cutoff_fn="cosine") -> "ANIAngular":
        return cls.cover_linearly(start, cutoff, eta, zeta, num_shifts, num_sections, cutoff_fn)


class _AEVBackwardFunction(torch.autograd.Function):
    """grad_aev -> grad_coords = J^T grad_aev as a differentiable function of grad_aev: its own backward is the
    forward-mode product J u (anihip_aev_jvp), the reference's cuaev double backward (CuaevDoubleAutograd,
    csrc/cuaev.cpp:141-186, csrc/aev.cu:1986-2015) that training on forces needs.  Like the reference it returns the
    derivative with respect to grad_aev only (no third order, no second-order term for the coordinates)."""

    @staticmethod
    def forward(ctx, grad_aev: Tensor, eng, nbrs, species32: Tensor) -> Tensor:
        g = grad_aev.detach().to(torch.float32).contiguous()
        ctx.eng, ctx.nbrs, ctx.species32 = eng, nbrs, species32
        ctx.g_dtype, ctx.g_shape = grad_aev.dtype, grad_aev.shape
        C, A = species32.shape
        return eng.backward(species32, nbrs, g).view(C, A, 3)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_force: Tensor):
        jt = ctx.eng.jvp(ctx.species32, ctx.nbrs, grad_force.contiguous())
        return jt.view(ctx.g_shape).to(ctx.g_dtype), None, None, None


class _AEVFunction(torch.autograd.Function):
    """coords -> aevs with the analytic HIP backward (the role of CuaevAutograd, csrc/cuaev.cpp:120-139)."""

    @staticmethod
    def forward(ctx, coords: Tensor, species32: Tensor, cell, pbc, computer: "AEVComputer") -> Tensor:
        eng = computer.engine()
        c32 = coords.detach().to(torch.float32).contiguous()
        nbrs = computer.neighbor_rows(species32, c32, cell, pbc)
        aev = eng.forward(species32, nbrs)
        ctx.eng, ctx.nbrs, ctx.species32 = eng, nbrs, species32
        ctx.in_dtype = coords.dtype
        computer._last_neighbors = nbrs
        return aev.view(species32.shape[0], species32.shape[1], eng.L).to(coords.dtype)

    @staticmethod
    def backward(ctx, grad_aev: Tensor):
        # (through a Function of its own so that create_graph=True can differentiate the forces once more)
        gc = _AEVBackwardFunction.apply(grad_aev, ctx.eng, ctx.nbrs, ctx.species32)
        return gc.to(ctx.in_dtype), None, None, None, None


class _AEVFromRowsFunction(torch.autograd.Function):
    """coords -> aevs for externally supplied neighbor rows (the role of cuaev::run_with_half_nbrlist,
    csrc/cuaev.cpp:205-224): the gradient flows to coords through the analytic HIP backward, exactly like the
    native operator, whose displacement inputs are not differentiated either."""

    @staticmethod
    def forward(ctx, coords: Tensor, species32: Tensor, nbrs: NeighborRows, computer: "AEVComputer") -> Tensor:
        eng = computer.engine()
        aev = eng.forward(species32, nbrs)
        ctx.eng, ctx.nbrs, ctx.species32 = eng, nbrs, species32
        ctx.in_dtype = coords.dtype
        computer._last_neighbors = nbrs
        return aev.view(species32.shape[0], species32.shape[1], eng.L).to(coords.dtype)

    @staticmethod
    def backward(ctx, grad_aev: Tensor):
        gc = _AEVBackwardFunction.apply(grad_aev, ctx.eng, ctx.nbrs, ctx.species32)
        return gc.to(ctx.in_dtype), None, None, None


class AEVComputer(torch.nn.Module):
    """Atomic environment vectors [C, A, S*16 + S(S+1)/2*32] on the MI355X engine.

    Arithmetic: the kernels compute in fp32 (binning and energy sums in fp64).  float64 coordinates are ACCEPTED like the
    reference's (its pyaev and cuAEV run in the input dtype, csrc/aev.cu:1742-1746) but are converted to fp32 on the way in,
    and the AEVs / gradients are cast back to float64 on the way out: fp64 in the interface, fp32-class accuracy in the
    numbers (AEV within ~3e-6 of the fp64 reference).  There is no fp64 kernel variant; the fp64 statement of the path is
    the test oracle (oracle/ani_oracle.c)."""

    def __init__(self, consts: AEVConstants, neighborlist: str = "auto", row_capacity: int = 128,
                 strategy: str = "hip", cutoff_fn: tp.Optional[str] = None, skin: float = 1.0) -> None:
        super().__init__()
        if strategy not in ("hip", "auto", "pyaev", "cuaev", "cuaev-fused", "cuaev-interface"):
            # the reference raises ValueError for unknown strategies (aev/_computer.py:127-128); its own strategy
            # names are accepted and all mean the native HIP path here
            raise ValueError(f"Unsupported strategy {strategy!r}: torchani_amd only has the native 'hip' path")
        # one cutoff function for both terms like the native strategies of the reference (aev/_computer.py:91-98);
        # an explicit argument overrides the one carried by the constants
        self.cutoff_fn = cutoff_kernel_name(consts.cutoff_fn if cutoff_fn is None else cutoff_fn)
        if self.cutoff_fn not in ("cosine", "smooth"):
            raise ValueError(f"Unsupported cutoff function {self.cutoff_fn!r}: the HIP kernels implement 'cosine' "
                             "(CutoffCosine) and 'smooth' (CutoffSmooth, order 2)")
        modes = {"auto": "auto", "all_pairs": "batch", "cell_list": "cell", "batch": "batch", "cell": "cell",
                 "adaptive": "auto", "base": "auto", "fast_cell_list": "cell", "verlet_cell_list": "cell",
                 "verlet": "auto"}
        if neighborlist not in modes:
            raise ValueError(f"Unsupported neighborlist {neighborlist!r}")   # neighbors.py:899-914
        self.neighbor_mode = modes[neighborlist]
        # Verlet-skin reuse of the pair search (VerletCellList, neighbors.py:759-884); "verlet" = the same on top of
        # the batched builder for several molecules
        self.verlet: tp.Optional[VerletRows] = VerletRows(skin) if neighborlist.startswith("verlet") else None
        self.row_capacity = int(row_capacity)
        self.num_species = consts.num_species
        self.radial = _RadialTerms(consts.EtaR, consts.ShfR, consts.Rcr)
        self.angular = _AngularTerms(consts.EtaA, consts.Zeta, consts.ShfA, consts.ShfZ, consts.Rca)
        self.register_buffer("triu_index", self._calculate_triu_index(consts.num_species))
        self._engine: tp.Optional[AevEngine] = None
        self._engine_key: tp.Optional[tuple] = None
        self._last_neighbors: tp.Optional[NeighborRows] = None
        self.strategy = "hip"
        # forward() reads the builder's status word after queueing its kernels (one host sync per call) and raises on
        # neighbor-row overflow; set False inside latency-critical loops that check last_neighbors() themselves
        self.check_overflow = True

    # aev/_computer.py:183-191
    @staticmethod
    def _calculate_triu_index(num_species: int) -> Tensor:
        s1, s2 = torch.triu_indices(num_species, num_species).unbind(0)
        pair_index = torch.arange(s1.shape[0], dtype=torch.long)
        ret = torch.zeros(num_species, num_species, dtype=torch.long)
        ret[s1, s2] = pair_index
        ret[s2, s1] = pair_index
        return ret

    @classmethod
    def like_2x(cls, num_species: int = 7, cutoff_fn: str = "cosine", **kw) -> "AEVComputer":
        return cls(aev_constants_2x(num_species, cutoff_fn), **kw)

    @classmethod
    def like_1x(cls, num_species: int = 4, cutoff_fn: str = "cosine", **kw) -> "AEVComputer":
        return cls(aev_constants_1x(num_species, cutoff_fn), **kw)

    @classmethod
    def from_terms(cls, radial: _RadialTerms, angular: _AngularTerms, num_species: int, cutoff_fn="cosine",
                   **kw) -> "AEVComputer":
        """From ANIRadial / ANIAngular objects (the reference's AEVComputer(radial=, angular=, num_species=, cutoff_fn=),
        aev/_computer.py:73-129): their hyper-parameters become the engine's table; one envelope for both terms."""
        if angular.cutoff > radial.cutoff:
            raise ValueError("Angular cutoff must be smaller or equal to radial cutoff")
        if angular.cutoff <= 0 or radial.cutoff <= 0:
            raise ValueError("Cutoffs must be strictly positive")
        return cls(AEVConstants(num_species, radial.cutoff, angular.cutoff, float(radial.eta.item()),
                                tuple(radial.shifts.double().tolist()), float(angular.eta.item()),
                                float(angular.zeta.item()), tuple(angular.shifts.double().tolist()),
                                tuple(angular.sections.double().tolist()), cutoff_fn), **kw)

    @classmethod
    def from_constants(cls, radial_cutoff: float, angular_cutoff: float, radial_eta: float,
                       radial_shifts: tp.Sequence[float], angular_eta: float, angular_zeta: float,
                       angular_shifts: tp.Sequence[float], sections: tp.Sequence[float], num_species: int,
                       cutoff_fn: str = "cosine", **kw) -> "AEVComputer":
        # aev/_computer.py:602-666
        return cls(AEVConstants(num_species, radial_cutoff, angular_cutoff, radial_eta, tuple(radial_shifts),
                                angular_eta, angular_zeta, tuple(angular_shifts), tuple(sections), cutoff_fn), **kw)

    # ---- derived sizes (aev/_computer.py:61-71,131-149) ----
    @property
    def num_species_pairs(self) -> int:
        return self.num_species * (self.num_species + 1) // 2

    @property
    def radial_len(self) -> int:
        return self.num_species * self.radial.shifts.numel()

    @property
    def angular_len(self) -> int:
        return self.num_species_pairs * self.angular.shifts.numel() * self.angular.sections.numel()

    @property
    def out_dim(self) -> int:
        return self.radial_len + self.angular_len

    def constants(self) -> AEVConstants:
        """Current hyper-parameters, read back from the (possibly state-dict-loaded) fp32 buffers."""
        r, a = self.radial, self.angular
        return AEVConstants(
            self.num_species, r.cutoff, a.cutoff, float(r.eta.item()),
            tuple(float(x) for x in r.shifts.tolist()), float(a.eta.item()), float(a.zeta.item()),
            tuple(float(x) for x in a.shifts.tolist()), tuple(float(x) for x in a.sections.tolist()),
            self.cutoff_fn)

    def engine(self) -> AevEngine:
        key = (self.radial.eta._version, self.radial.shifts._version, self.angular.eta._version,
               self.angular.zeta._version, self.angular.shifts._version, self.angular.sections._version,
               self.radial.cutoff, self.angular.cutoff, self.cutoff_fn)
        if self._engine is None or self._engine_key != key:
            self._engine = AevEngine(self.constants())
            self._engine_key = key
        return self._engine

    def neighbor_rows(self, species32: Tensor, c32: Tensor, cell: tp.Optional[Tensor] = None, pbc=None,
                      lo: int = 0, hi: tp.Optional[int] = None) -> NeighborRows:
        """Neighbor rows of the central atoms lo..hi for this computer's neighborlist setting (pair search, or
        the Verlet-skin refresh of an earlier one)."""
        eng = self.engine()
        hi = species32.numel() if hi is None else hi
        if self.verlet is not None:
            mode = self.neighbor_mode
            if mode == "auto":
                mode = "cell" if (species32.shape[0] == 1 and species32.shape[1] > 512) else "batch"
            return self.verlet.rows(eng, species32, c32, cell, pbc, lo, hi, mode, self.row_capacity)
        return eng.neighbors(species32, c32, cell, pbc, lo=lo, hi=hi, mode=self.neighbor_mode,
                             row_cap=self.row_capacity)

    def compute_from_neighbors(self, elem_idxs: Tensor, coords: Tensor, neighbors: Neighbors) -> Tensor:
        """AEVs from the result of an external neighbor-list calculation (aev/_computer.py:251-272): any
        3-tuple (indices [2, P], distances [P], diff_vectors [P, 3]) in the reference's convention."""
        if not coords.is_cuda:
            raise ValueError("torchani_amd's AEVComputer needs tensors on a ROCm device (no CPU fallback)")
        if elem_idxs.dim() != 2 or coords.shape != (elem_idxs.shape[0], elem_idxs.shape[1], 3):
            raise ValueError("expected elem_idxs [C, A] and coords [C, A, 3]")
        indices, _, diff_vectors = neighbors
        species32 = elem_idxs.to(torch.int32).contiguous()
        nbrs = self.engine().rows_from_half(species32, indices.to(coords.device), diff_vectors.to(coords.device),
                                            row_cap=self.row_capacity)
        return _AEVFromRowsFunction.apply(coords, species32, nbrs, self)

    def compute_from_full_nbrlist(self, elem_idxs: Tensor, coords: Tensor, ilist_unique: Tensor, jlist: Tensor,
                                  numneigh: Tensor) -> Tensor:
        """AEVs from a LAMMPS-style full neighbor list over local + ghost atoms (the reference's
        _compute_cuaev_with_full_nbrlist, aev/_computer.py:420-438): one molecule, atoms that are not in
        ilist_unique get zero rows."""
        if not coords.is_cuda:
            raise ValueError("torchani_amd's AEVComputer needs tensors on a ROCm device (no CPU fallback)")
        if coords.shape[0] != 1:
            raise ValueError("the full-neighborlist entry point doesn't support batches")
        species32 = elem_idxs.to(torch.int32).contiguous()
        c32 = coords.detach().to(torch.float32).contiguous()
        nbrs = self.engine().rows_from_full(species32, c32, ilist_unique, jlist, numneigh, row_cap=self.row_capacity)
        return _AEVFromRowsFunction.apply(coords, species32, nbrs, self)

    _compute_cuaev_with_full_nbrlist = compute_from_full_nbrlist   # the reference's (private) name

    def set_strategy(self, strategy: str) -> None:
        # (aev/_computer.py:131-149: the reference's names are accepted like in the constructor, all mean the HIP path)
        if strategy not in ("hip", "auto", "pyaev", "cuaev", "cuaev-fused", "cuaev-interface"):
            raise ValueError(f"Unsupported strategy {strategy!r}")

    def forward(self, elem_idxs, coords: tp.Optional[Tensor] = None, cell: tp.Optional[Tensor] = None,
                pbc: tp.Optional[Tensor] = None):
        """aevs = aevc(elem_idxs, coords, cell=None, pbc=None); the legacy tuple form
        ``_, aevs = aevc((idxs, coords), cell, pbc)`` is accepted with a warning
        (aev/_computer.py:211-222)."""
        if isinstance(elem_idxs, tuple):
            warnings.warn("The tuple call signature is deprecated; use aevc(elem_idxs, coords, cell, pbc)")
            idxs, crd = elem_idxs
            # legacy call: (species, coords), cell, pbc were positional
            cell, pbc = (coords if coords is not None else cell), (cell if coords is not None else pbc)
            return SpeciesAEV(idxs, self.forward(idxs, crd, cell, pbc))
        assert coords is not None
        if not coords.is_cuda:
            raise ValueError("torchani_amd's AEVComputer needs tensors on a ROCm device (no CPU fallback)")
        if elem_idxs.dim() != 2 or coords.shape != (elem_idxs.shape[0], elem_idxs.shape[1], 3):
            raise ValueError("expected elem_idxs [C, A] and coords [C, A, 3]")
        if (cell is None) != (pbc is None):
            raise ValueError("cell and pbc must be given together")
        pbc_t = None if pbc is None else tuple(bool(b) for b in pbc.tolist())
        species32 = elem_idxs.to(torch.int32).contiguous()
        out = _AEVFunction.apply(coords, species32, cell, pbc_t, self)
        if self.check_overflow and not torch.cuda.is_current_stream_capturing() and self._last_neighbors.overflowed():
            # a row over capacity was zeroed by the builder: never return such AEVs silently (the reference
            # asserts on the device, csrc/aev.cu:229); one retry at the largest row capacity first
            if self.row_capacity < _MAX_RAD:
                warnings.warn(f"neighbor rows overflowed row_capacity={self.row_capacity}: retrying with {_MAX_RAD}")
                self.row_capacity = _MAX_RAD
                out = _AEVFunction.apply(coords, species32, cell, pbc_t, self)
            self._last_neighbors.raise_on_overflow()
        return out

    def last_neighbors(self) -> tp.Optional[NeighborRows]:
        return self._last_neighbors

    def extra_repr(self) -> str:
        return (f"num_species={self.num_species}, out_dim={self.out_dim}, strategy=hip, "
                f"neighborlist={self.neighbor_mode}, row_capacity={self.row_capacity}")


_cell_list_engines: tp.Dict[float, AevEngine] = {}


def cell_list(cutoff: float, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None,
              pbc: tp.Optional[Tensor] = None) -> Neighbors:
    """Half neighbor list of ONE system from the engine's cell-list rows, in the reference's format -- the counterpart of
    ``torch.ops.cell_list.cell_list(cutoff, species, coords, cell, pbc)`` behind ``FastCellList`` (csrc/cell_list.cpp:
    342-354, neighbors.py:278-294): Neighbors(indices [2, P] int64, distances [P], diff_vectors [P, 3]) with every pair
    inside ``cutoff`` once, dummy atoms (species -1) dropped, diff = r[indices[0]] - r[indices[1]] (+ image shift).
    Not differentiable (the reference recomputes diff from the coordinates for autograd, neighbors.py:105-112)."""
    from .constants import aev_constants_2x
    from .engine import rows_to_half

    if not coords.is_cuda:
        raise ValueError("torchani_amd's neighbor builders need tensors on a ROCm device (no CPU fallback)")
    if species.dim() != 2 or species.shape[0] != 1 or coords.shape != (1, species.shape[1], 3):
        raise ValueError("cell_list handles one system at a time: species [1, A], coords [1, A, 3] (neighbors.py:373-381)")
    eng = _cell_list_engines.get(float(cutoff))
    if eng is None:
        eng = _cell_list_engines[float(cutoff)] = AevEngine(aev_constants_2x(7)._replace(Rcr=float(cutoff), Rca=1e-3))
    pbc_t = None if (pbc is None or cell is None) else tuple(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc))
    sp32 = species.clamp(min=-1, max=0).to(torch.int32).contiguous()   # (only "dummy or not" matters for the pair search)
    rows = eng.neighbors(sp32, coords.detach().to(torch.float32).contiguous(), cell, pbc_t, mode="cell",
                         row_cap=_MAX_RAD)
    rows.raise_on_overflow()
    idx, dist, diff = rows_to_half(rows, species.shape[1])
    return Neighbors(idx, dist.to(coords.dtype), diff.to(coords.dtype))
