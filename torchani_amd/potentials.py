"""Pair potentials on the engine's neighbor rows: the xTB repulsion and DFT-D3(BJ) dispersion terms of the reference's
ANI-2xr / ANI-2dr models, and the other closed-form pair potentials of torchani.potentials (RepulsionZBL, LennardJones /
RepulsionLJ / DispersionLJ, FixedCoulomb, FixedMNOK).

Mirrors torchani/potentials/xtb.py:17-77 (RepulsionXTB: constructor, buffers ``y_ab`` / ``sqrt_alpha_ab`` / ``k_rep_ab``,
pair energies), torchani/potentials/dftd3.py:44-330 (BeckeJohnsonDamp, TwoBodyDispersionD3: constructor,
``from_functional``, coordination numbers, C6 interpolation, pair energies) and the shared machinery of
torchani/potentials/core.py:103-207 (cutoff envelope, ``atomic`` halves, sum per molecule).  The arithmetic runs in
libanihip (anihip_pair_xtb_repulsion, anihip_pair_d3, csrc/pair.hip) on neighbor rows of the engine; there is no eager
fallback.

(The GELU / bias-free networks of those models run through the fused network kernel, torchani_amd.models.ANI2xr /
ANI2dr.)
"""
from __future__ import annotations

import ctypes as C
import math
import os
import typing as tp

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .constants import ATOMIC_NUMBER
from .engine import AevEngine, NeighborRows, _ptr, _require_cuda, _stream

# resources/atomic_constants.json "xtb_repulsion_alpha" / "xtb_repulsion_yeff" (Grimme et al., GFN2-xTB,
# https://pubs.acs.org/doi/10.1021/acs.jctc.8b01176), elements up to Kr
XTB_REPULSION: tp.Dict[str, tp.Tuple[float, float]] = {
    "H": (2.213717, 1.105388), "He": (3.60467, 1.094283), "Li": (0.475307, 1.289367), "Be": (0.939696, 4.221216),
    "B": (1.373856, 7.192431), "C": (1.247655, 4.231078), "N": (1.682689, 5.242592), "O": (2.165712, 5.784415),
    "F": (2.421394, 7.021486), "Ne": (3.318479, 11.041068), "Na": (0.572728, 5.244917), "Mg": (0.917975, 18.083164),
    "Al": (0.876623, 17.867328), "Si": (1.187323, 40.001111), "P": (1.143343, 19.683502), "S": (1.214553, 14.99509),
    "Cl": (1.577144, 17.353134), "Ar": (0.896198, 7.266606), "K": (0.482206, 10.439482), "Ca": (0.683051, 14.786701),
    "Br": (1.296174, 32.845361), "Kr": (0.908074, 17.363803),
}


class _PairEnergy(torch.autograd.Function):
    """coords -> molecular pair energies [C] (float64) with the kernel's analytic gradient."""

    @staticmethod
    def forward(ctx, coords: Tensor, pot: "RepulsionXTB", species32: Tensor, nbrs: NeighborRows) -> Tensor:
        Cn, A = species32.shape
        atomic = torch.zeros(Cn * A, dtype=torch.float32, device=coords.device)
        grad = torch.zeros((Cn * A, 3), dtype=torch.float32, device=coords.device)
        pot.accumulate(species32, nbrs, atomic, grad)
        ctx.save_for_backward(grad)
        ctx.shape, ctx.dtype = coords.shape, coords.dtype
        return atomic.view(Cn, A).to(torch.float64).sum(dim=1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g: Tensor):
        (grad,) = ctx.saved_tensors
        Cn, A = ctx.shape[0], ctx.shape[1]
        return (grad.view(Cn, A, 3) * g.view(Cn, 1, 1).to(grad.dtype)).to(ctx.dtype), None, None, None


class _Standalone:
    """``potential(species, coords, cell, pbc, atomic=False, atomic_nums_input=True)`` without a model around it
    (core.py:37-67 Potential.forward): the potential builds its own neighbor rows with its own cutoff and evaluates
    itself on them.  Molecular energies [C] (float64, differentiable with respect to coords) or per-atom halves [C, A]."""

    def _standalone_rows(self, species32: Tensor, coords: Tensor, cell, pbc) -> NeighborRows:
        from .constants import aev_constants_2x

        rc = min(self.cutoff, 1.0e3)
        if self._own_engine is None or abs(self._own_engine.consts.Rcr - rc) > 1e-9:
            self._own_engine = AevEngine(aev_constants_2x(len(self.symbols))._replace(Rcr=rc, Rca=1e-3))
        pbc_t = None if pbc is None else tuple(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc))
        mode = "cell" if (species32.shape[0] == 1 and species32.shape[1] > 512) else "batch"
        rows = self._own_engine.neighbors(species32, coords.detach().to(torch.float32).contiguous(), cell, pbc_t, mode=mode,
                                          row_cap=_lib.MAX_RAD)
        self._last_rows = rows
        return rows

    def _to_elem_idxs(self, species: Tensor, atomic_nums_input: bool) -> Tensor:
        """Atomic numbers -> this potential's element indices (padding -1 stays); unknown elements raise."""
        if not atomic_nums_input:
            return species
        lut = torch.full((120,), -1, dtype=torch.long)
        lut[self.atomic_numbers.cpu()] = torch.arange(len(self.symbols))
        elem = lut.to(species.device)[species.clamp(min=-1)]     # (-1 indexes the last, unused slot)
        if (elem[species != -1] == -1).any():
            raise ValueError(f"Unsupported element in {torch.unique(species).tolist()}: this potential knows {self.symbols}")
        return elem

    def forward(self, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None, pbc=None, atomic: bool = False,
                ensemble_values: bool = False, atomic_nums_input: bool = True) -> Tensor:
        if not coords.is_cuda:
            raise ValueError("torchani_amd's pair potentials need tensors on a ROCm device (no CPU fallback)")
        if species.dim() != 2 or coords.shape != (species.shape[0], species.shape[1], 3):
            raise ValueError("expected species [C, A] and coords [C, A, 3]")
        species32 = self._to_elem_idxs(species, atomic_nums_input).to(torch.int32).contiguous()
        rows = self._standalone_rows(species32, coords, cell, pbc)
        if atomic:
            a = torch.zeros(species32.numel(), dtype=torch.float32, device=coords.device)
            self.accumulate(species32, rows, a, None)
            e = a.view(species32.shape)
        else:
            e = self.compute_from_rows(species32, coords, rows)
        if not torch.cuda.is_current_stream_capturing() and rows.overflowed():   # (the builder zeroed the row)
            raise RuntimeError(f"{type(self).__name__}: an atom has more than {_lib.MAX_RAD} neighbors inside the cutoff "
                               f"({self.cutoff} A); rows hold at most {_lib.MAX_RAD}")
        return e.unsqueeze(0) if ensemble_values else e


class _AnalyticPair(_Standalone, torch.nn.Module):
    """Shared part of the closed-form pair potentials (core.py:103-207 BasePairPotential): a [8, 8, 4] device table of
    per-element-pair constants evaluated by anihip_pair_analytic (kind = ANIHIP_PAIR_*)."""

    kind = _lib.PAIR_XTB
    clamp_distances = True   # (core.py:138-139; FixedMNOK does not clamp)

    def _init_common(self, symbols: tp.Sequence[str], cutoff: float, cutoff_fn) -> None:
        from .constants import cutoff_kernel_name as kernel_name

        cutoff_fn = kernel_name(cutoff_fn)
        if cutoff_fn not in _lib.CUTOFF_KINDS:
            raise ValueError(f"Unsupported cutoff function {cutoff_fn!r}: the HIP kernels have {sorted(_lib.CUTOFF_KINDS)}")
        if len(symbols) > 7:
            raise ValueError("at most 7 elements (the species field of a neighbor row)")
        self.symbols = tuple(symbols)
        self.register_buffer("atomic_numbers", torch.tensor([ATOMIC_NUMBER.get(s, 0) for s in symbols]))
        self.cutoff = float(cutoff)
        self.cutoff_fn = cutoff_fn
        self._enabled = True
        self._table: tp.Optional[Tensor] = None
        self._own_engine: tp.Optional[AevEngine] = None

    def _elem_seq(self, name: str, seq: tp.Sequence[float], default: tp.Optional[tp.Callable[[str], float]] = None
                  ) -> tp.List[float]:
        # _core.py:32-54 _validate_elem_seq
        if not seq and default is not None:
            seq = [float(default(s)) for s in self.symbols]
        if not all(isinstance(v, float) for v in seq):
            raise ValueError(f"Some values in {name} are not floats")
        if len(seq) != len(self.symbols):
            raise ValueError(f"{name} and symbols should have the same len")
        return list(seq)

    def _pair_constants(self) -> Tensor:
        """[S, S, 4] host tensor of the kernel's per-pair constants."""
        raise NotImplementedError

    def _extra(self) -> tp.Optional[np.ndarray]:
        return None

    def table(self, device: torch.device) -> Tensor:
        if self._table is None or self._table.device != device:
            S = len(self.symbols)
            t = torch.zeros((8, 8, 4), dtype=torch.float32)
            t[:S, :S] = self._pair_constants().to(torch.float32)
            self._table = t.to(device).contiguous()
        return self._table

    def rows_cutoff(self, rows_rcr: float) -> float:
        """Cutoff to evaluate with on rows built with radial cutoff rows_rcr (inf = everything the rows hold)."""
        if self.cutoff > rows_rcr + 1e-6 and not math.isinf(self.cutoff):
            raise ValueError(f"pair cutoff {self.cutoff} exceeds the neighbor rows' cutoff {rows_rcr}")
        return self.cutoff

    def accumulate(self, species32: Tensor, nbrs: NeighborRows, atomic_e: tp.Optional[Tensor],
                   grad_coords: tp.Optional[Tensor], virial: tp.Optional[Tensor] = None,
                   cutoff: tp.Optional[float] = None) -> None:
        """atomic_e [N] += pair halves, grad_coords [N, 3] += gradient, virial [3, 3] += for the central atoms of nbrs."""
        _require_cuda(species32, atomic_e, grad_coords, virial)
        cut = self.cutoff if cutoff is None else cutoff
        if math.isinf(cut):
            cut = 1e30   # the rows decide (with the envelope == 1 up to rounding at finite distances)
        flags = (0 if nbrs.symmetric else _lib.PAIR_PUSH) | (0 if self.clamp_distances else _lib.PAIR_NO_CLAMP)
        ex = self._extra()
        _lib.check(_lib.lib().anihip_pair_analytic(
            _stream(), self.kind, species32.numel(), nbrs.lo, nbrs.hi, _ptr(species32), _ptr(nbrs.meta), _ptr(nbrs.ent),
            _ptr(self.table(species32.device)), None if ex is None else ex.ctypes.data, float(cut),
            _lib.CUTOFF_KINDS[self.cutoff_fn], flags, _ptr(atomic_e), _ptr(grad_coords), _ptr(virial)))

    def compute_from_rows(self, species32: Tensor, coords: Tensor, nbrs: NeighborRows) -> Tensor:
        """Molecular energies [C] (float64), differentiable with respect to coords."""
        return _PairEnergy.apply(coords, self, species32, nbrs)

    def extra_repr(self) -> str:
        return f"symbols={self.symbols}, cutoff={self.cutoff}, cutoff_fn={self.cutoff_fn}"


class RepulsionXTB(_AnalyticPair):
    """xTB repulsion pair potential (potentials/xtb.py:17-77).  ``krep_hydrogen`` applies to H-H pairs only."""

    kind = _lib.PAIR_XTB

    def __init__(self, symbols: tp.Sequence[str], krep_hydrogen: float = 1.0, krep: float = 1.5,
                 alpha: tp.Sequence[float] = (), yeff: tp.Sequence[float] = (), *, cutoff: float = math.inf,
                 cutoff_fn: str = "smooth") -> None:
        super().__init__()
        self._init_common(symbols, cutoff, cutoff_fn)
        for name, seq in (("alpha", alpha), ("yeff", yeff)):
            if seq and len(seq) != len(symbols):
                raise ValueError(f"len({name}), if provided, must match len(symbols)")   # core.py _validate_elem_seq
        missing = [s for s in symbols if s not in XTB_REPULSION and not (alpha and yeff)]
        if missing:
            raise ValueError(f"no xTB repulsion constants for {missing}: pass alpha and yeff")
        a = torch.tensor(list(alpha) if alpha else [XTB_REPULSION[s][0] for s in symbols], dtype=torch.float32)
        y = torch.tensor(list(yeff) if yeff else [XTB_REPULSION[s][1] for s in symbols], dtype=torch.float32)
        k = torch.full((len(symbols), len(symbols)), float(krep))
        if "H" in self.symbols:
            h = self.symbols.index("H")
            k[h, h] = float(krep_hydrogen)
        self.register_buffer("y_ab", torch.outer(y, y))
        self.register_buffer("sqrt_alpha_ab", torch.outer(a, a).sqrt())
        self.register_buffer("k_rep_ab", k)

    def _pair_constants(self) -> Tensor:
        z = torch.zeros_like(self.y_ab.cpu())
        return torch.stack([self.y_ab.cpu(), self.sqrt_alpha_ab.cpu(), self.k_rep_ab.cpu(), z], dim=-1)


class RepulsionZBL(_AnalyticPair):
    """Ziegler-Biersack-Littmark screened nuclear repulsion (potentials/zbl.py:10-81)."""

    kind = _lib.PAIR_ZBL

    def __init__(self, symbols: tp.Sequence[str], k: float = 0.8853, screen_coeffs: tp.Sequence[float] = (),
                 screen_exponents: tp.Sequence[float] = (), eff_exponent: float = 0.23,
                 eff_atomic_nums: tp.Sequence[float] = (), *, cutoff: float = math.inf, cutoff_fn: str = "smooth") -> None:
        super().__init__()
        self._init_common(symbols, cutoff, cutoff_fn)
        z = self._elem_seq("eff_atomic_nums", eff_atomic_nums, lambda s: float(ATOMIC_NUMBER[s]))
        if len(screen_exponents) != len(screen_coeffs):
            raise ValueError("screen_exponents and screen_coeffs must have the same len")
        c = list(screen_coeffs) if screen_coeffs else [0.18175, 0.50986, 0.28022, 0.02817]
        b = list(screen_exponents) if screen_exponents else [3.19980, 0.94229, 0.40290, 0.20162]
        if not math.isclose(sum(c), 1.0):
            raise ValueError("Screen coeffs must sum to 1")
        if len(c) > 4:
            raise ValueError("the kernel holds up to 4 screening terms")
        self.register_buffer("_eff_atomic_nums", torch.tensor(z), persistent=False)
        self._k, self._kz = float(k), float(eff_exponent)
        self._screen = np.asarray(c + [0.0] * (4 - len(c)) + b + [0.0] * (4 - len(b)), dtype=np.float32)

    def _pair_constants(self) -> Tensor:
        z = self._eff_atomic_nums.cpu().to(torch.float64)
        zz = torch.outer(z, z)
        s = (z.pow(self._kz).unsqueeze(1) + z.pow(self._kz).unsqueeze(0)) / self._k
        o = torch.zeros_like(zz)
        return torch.stack([zz, s, o, o], dim=-1)

    def _extra(self) -> np.ndarray:
        return self._screen


_LJ_EPS = 0.1 / 627.5094738898777   # Hartree (lj.py:14: 0.1 kcal/mol)
_LJ_SIGMA = 1.5                      # Angstrom


class _LJ(_AnalyticPair):
    """Lennard-Jones terms with Lorentz-Berthelot combination (potentials/lj.py:42-108): sigma in Angstrom, eps in
    Hartree, defaults sigma = 1.5, eps = 0.1 kcal/mol for every element."""

    kind = _lib.PAIR_LJ
    c12, c6 = 1.0, -1.0

    def __init__(self, symbols: tp.Sequence[str], eps: tp.Sequence[float] = (), sigma: tp.Sequence[float] = (), *,
                 cutoff: float = math.inf, cutoff_fn: str = "smooth") -> None:
        super().__init__()
        self._init_common(symbols, cutoff, cutoff_fn)
        self.register_buffer("_eps", torch.tensor(self._elem_seq("eps", eps, lambda s: _LJ_EPS)), persistent=False)
        self.register_buffer("_sigma", torch.tensor(self._elem_seq("sigma", sigma, lambda s: _LJ_SIGMA)), persistent=False)

    def _pair_constants(self) -> Tensor:
        e, sg = self._eps.cpu().to(torch.float64), self._sigma.cpu().to(torch.float64)
        eps_ab = torch.sqrt(torch.outer(e, e))                      # lj.py:84-85
        sig_ab = (sg.unsqueeze(1) + sg.unsqueeze(0)) / 2            # lj.py:86-87
        return torch.stack([4 * eps_ab, sig_ab, torch.full_like(eps_ab, self.c12), torch.full_like(eps_ab, self.c6)], dim=-1)


class LennardJones(_LJ):
    """4 eps ((sigma / r)^12 - (sigma / r)^6) (lj.py:102-108)."""


class RepulsionLJ(_LJ):
    """4 eps (sigma / r)^12 (lj.py:95-101)."""
    c12, c6 = 1.0, 0.0


class DispersionLJ(_LJ):
    """-4 eps (sigma / r)^6 (lj.py:88-94)."""
    c12, c6 = 0.0, -1.0


class FixedCoulomb(_AnalyticPair):
    """Coulomb energy of fixed per-element charges (potentials/fixed_coulomb.py:8-31)."""

    kind = _lib.PAIR_COULOMB

    def __init__(self, symbols: tp.Sequence[str], dielectric: float = 1.0, charges: tp.Sequence[float] = (), *,
                 cutoff: float = math.inf, cutoff_fn: str = "smooth") -> None:
        super().__init__()
        self._init_common(symbols, cutoff, cutoff_fn)
        self._dielectric = float(dielectric)
        self.register_buffer("_charges", torch.tensor(self._elem_seq("charges", charges)), persistent=False)

    def _pair_constants(self) -> Tensor:
        q = self._charges.cpu().to(torch.float64)
        qq = torch.outer(q, q) / self._dielectric
        o = torch.zeros_like(qq)
        return torch.stack([qq, o, o, o], dim=-1)


class FixedMNOK(_AnalyticPair):
    """Mataga-Nishimoto-Ohno-Klopman damped Coulomb energy of fixed charges (potentials/fixed_coulomb.py:34-75):
    q_a q_b / sqrt(d^2 + (2 / (eta_a + eta_b))^2); like the reference's, it neither clamps distances nor applies the
    dielectric constant."""

    kind = _lib.PAIR_COULOMB
    clamp_distances = False

    def __init__(self, symbols: tp.Sequence[str], dielectric: float = 1.0, charges: tp.Sequence[float] = (),
                 eta: tp.Sequence[float] = (), *, cutoff: float = math.inf, cutoff_fn: str = "smooth") -> None:
        super().__init__()
        self._init_common(symbols, cutoff, cutoff_fn)
        self._dielectric = float(dielectric)
        self.register_buffer("_charges", torch.tensor(self._elem_seq("charges", charges)), persistent=False)
        self.register_buffer("_eta", torch.tensor(self._elem_seq("eta", eta)), persistent=False)

    def _pair_constants(self) -> Tensor:
        q, eta = self._charges.cpu().to(torch.float64), self._eta.cpu().to(torch.float64)
        qq = torch.outer(q, q)
        inv_eta = 2 / (eta.unsqueeze(1) + eta.unsqueeze(0))          # fixed_coulomb.py:62-63
        o = torch.zeros_like(qq)
        return torch.stack([qq, inv_eta, o, o], dim=-1)


_D3_REFS: tp.Optional[tp.Dict[str, tp.Any]] = None


def d3_reference_data() -> tp.Dict[str, tp.Any]:
    """Grimme's D3 reference data for Z <= 18 (torchani_amd/data/d3_refs.npz, extracted from the reference's
    resources/c6.h5, atomic_constants.json and functional_d3bj_constants.json by tests/golden/gen_golden_d3.py)."""
    global _D3_REFS
    if _D3_REFS is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "d3_refs.npz")
        with np.load(path) as z:
            d = {k: z[k] for k in z.files}
        d["symbols"] = [str(s) for s in d["symbols"]]
        d["functionals"] = {str(n): tuple(float(v) for v in row)
                            for n, row in zip(d["functionals"], d["functional_s6_s8_a1_a2"])}
        _D3_REFS = d
    return _D3_REFS


class TwoBodyDispersionD3(_Standalone, torch.nn.Module):
    """Two-body DFT-D3 dispersion with Becke-Johnson damping (potentials/dftd3.py:113-330).

    ``sqrt_empirical_charge`` / ``covalent_radii`` (Angstrom) default to the tabulated values of the elements, like the
    reference's; the C6 reference table covers Z <= 18."""

    ANGSTROM_TO_BOHR = 1.8897261258369282   # torchani/units.py:41

    def __init__(self, symbols: tp.Sequence[str], s6: float, s8: float, damp_a1: float, damp_a2: float,
                 sqrt_empirical_charge: tp.Sequence[float] = (), covalent_radii: tp.Sequence[float] = (), *,
                 cutoff_fn: str = "smooth", cutoff: float = math.inf) -> None:
        super().__init__()
        from .constants import cutoff_kernel_name as kernel_name

        cutoff_fn = kernel_name(cutoff_fn)
        if cutoff_fn not in _lib.CUTOFF_KINDS:
            raise ValueError(f"Unsupported cutoff function {cutoff_fn!r}: the HIP kernels have {sorted(_lib.CUTOFF_KINDS)}")
        if len(symbols) > 7:
            raise ValueError("at most 7 elements (the species field of a neighbor row)")
        ref = d3_reference_data()
        self.symbols = tuple(symbols)
        unknown = [s for s in symbols if s not in ref["symbols"]]
        if unknown:
            raise ValueError(f"no D3 reference data for {unknown} (the shipped table covers {ref['symbols']})")
        z = [ref["symbols"].index(s) + 1 for s in symbols]
        for name, seq in (("sqrt_empirical_charge", sqrt_empirical_charge), ("covalent_radii", covalent_radii)):
            if seq and len(seq) != len(symbols):
                raise ValueError(f"len({name}), if provided, must match len(symbols)")   # core.py _validate_elem_seq
        sq = list(sqrt_empirical_charge) if sqrt_empirical_charge else [float(ref["sqrt_empirical_charge"][k]) for k in z]
        cov = list(covalent_radii) if covalent_radii else [float(ref["covalent_radius"][k]) for k in z]
        self._s6, self._s8, self._a1, self._a2 = float(s6), float(s8), float(damp_a1), float(damp_a2)
        zi = np.asarray(z)
        self.register_buffer("atomic_numbers", torch.tensor(z))
        self.register_buffer("precalc_coeff6", torch.from_numpy(ref["c6"][zi][:, zi].copy()))
        self.register_buffer("precalc_coordnums_a", torch.from_numpy(ref["cn_a"][zi][:, zi].copy()))
        self.register_buffer("precalc_coordnums_b", torch.from_numpy(ref["cn_b"][zi][:, zi].copy()))
        _sq = torch.tensor(sq, dtype=torch.float32)
        self.register_buffer("sqrt_charge_ab", torch.outer(_sq, _sq))
        self.register_buffer("covalent_radii", torch.tensor([self.ANGSTROM_TO_BOHR * r for r in cov], dtype=torch.float32))
        self._sqrt_q = [float(v) for v in sq]
        self._cov_bohr = [float(self.ANGSTROM_TO_BOHR * r) for r in cov]   # host copy: params() must not sync the device
        self._params: tp.Optional["_lib.D3Params"] = None
        self.cutoff = float(cutoff)
        self.cutoff_fn = cutoff_fn
        self._enabled = True
        self._table: tp.Optional[Tensor] = None
        self._own_engine: tp.Optional[AevEngine] = None
        self.needs_all_rows = True   # coordination numbers of every neighbor: rows of all atoms, not of a shard

    @classmethod
    def from_functional(cls, symbols: tp.Sequence[str], functional: str, *, cutoff_fn: str = "smooth",
                        cutoff: float = math.inf) -> "TwoBodyDispersionD3":
        fn = d3_reference_data()["functionals"]
        if functional.lower() not in fn:
            raise ValueError(f"no D3(BJ) constants for functional {functional!r}")
        s6, s8, a1, a2 = fn[functional.lower()]
        return cls(symbols, s6=s6, s8=s8, damp_a1=a1, damp_a2=a2, cutoff_fn=cutoff_fn, cutoff=cutoff)

    def table(self, device: torch.device) -> Tensor:
        """[8, 8, 25, 4] device table {c6 ref, cn_a ref, cn_b ref, -}: valid references first, their count in [.., 0, 3]
        (include/anihip.h)."""
        if self._table is None or self._table.device != device:
            S = len(self.symbols)
            t = torch.zeros((8, 8, 25, 4), dtype=torch.float32)
            c6 = self.precalc_coeff6.cpu().reshape(S, S, 25)
            ca = self.precalc_coordnums_a.cpu().reshape(S, S, 25)
            cb = self.precalc_coordnums_b.cpu().reshape(S, S, 25)
            for a in range(S):
                for b in range(S):
                    ok = torch.nonzero(c6[a, b] > 0.0).reshape(-1)   # (missing references are -1: dftd3.py:318-321)
                    t[a, b, :len(ok), 0], t[a, b, :len(ok), 1], t[a, b, :len(ok), 2] = c6[a, b, ok], ca[a, b, ok], cb[a, b, ok]
                    t[a, b, 0, 3] = float(len(ok))
            self._table = t.to(device).contiguous()
        return self._table

    def params(self) -> "_lib.D3Params":
        if self._params is None:   # built once from host copies (no device reads: legal during stream capture)
            p = _lib.D3Params()
            p.s6, p.s8, p.a1, p.a2 = self._s6, self._s8, self._a1, self._a2
            for k in range(8):
                p.cov_radius_bohr[k] = self._cov_bohr[k] if k < len(self.symbols) else 0.0
                p.sqrt_q[k] = self._sqrt_q[k] if k < len(self.symbols) else 0.0
            self._params = p
        return self._params

    def rows_cutoff(self, rows_rcr: float) -> float:
        if self.cutoff > rows_rcr + 1e-6 and not math.isinf(self.cutoff):
            raise ValueError(f"pair cutoff {self.cutoff} exceeds the neighbor rows' cutoff {rows_rcr}")
        return self.cutoff

    def accumulate(self, species32: Tensor, nbrs: NeighborRows, atomic_e: tp.Optional[Tensor],
                   grad_coords: tp.Optional[Tensor], virial: tp.Optional[Tensor] = None,
                   cutoff: tp.Optional[float] = None, lo: tp.Optional[int] = None, hi: tp.Optional[int] = None) -> None:
        """atomic_e [N] += pair halves, grad_coords [N, 3] += gradient, virial [3, 3] +=, for the central atoms
        lo .. hi (default: all).  nbrs must hold the rows of ALL atoms and be symmetric."""
        _require_cuda(species32, atomic_e, grad_coords, virial)
        n = species32.numel()
        if not nbrs.symmetric or nbrs.lo != 0 or nbrs.hi != n:
            raise ValueError("TwoBodyDispersionD3 needs symmetric neighbor rows of all atoms")
        cut = self.cutoff if cutoff is None else cutoff
        if math.isinf(cut):
            cut = 1e30
        dev = species32.device
        cn = torch.empty(n, dtype=torch.float32, device=dev)
        gcn = torch.empty(n, dtype=torch.float32, device=dev)
        p = self.params()
        _lib.check(_lib.lib().anihip_pair_d3(
            _stream(), n, 0 if lo is None else lo, n if hi is None else hi, _ptr(species32), _ptr(nbrs.meta),
            _ptr(nbrs.ent), _ptr(self.table(dev)), C.byref(p), float(cut), _lib.CUTOFF_KINDS[self.cutoff_fn], _ptr(cn),
            _ptr(gcn), _ptr(atomic_e), _ptr(grad_coords), _ptr(virial)))

    def compute_from_rows(self, species32: Tensor, coords: Tensor, nbrs: NeighborRows) -> Tensor:
        """Molecular energies [C] (float64), differentiable with respect to coords."""
        return _PairEnergy.apply(coords, self, species32, nbrs)

    def extra_repr(self) -> str:
        return (f"symbols={self.symbols}, s6={self._s6}, s8={self._s8}, a1={self._a1}, a2={self._a2}, "
                f"cutoff={self.cutoff}, cutoff_fn={self.cutoff_fn}")
