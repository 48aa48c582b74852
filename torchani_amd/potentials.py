"""Pair potentials on the engine's neighbor rows: the xTB repulsion term of the reference's ANI-2xr / ANI-2dr models.

Mirrors torchani/potentials/xtb.py:17-77 (RepulsionXTB: constructor, buffers ``y_ab`` / ``sqrt_alpha_ab`` / ``k_rep_ab``,
pair energies) and the shared machinery of torchani/potentials/core.py:103-207 (cutoff envelope, ``atomic`` halves, sum
per molecule).  The arithmetic runs in libanihip (anihip_pair_xtb_repulsion, csrc/pair.hip) on the same rows the AEV
kernels use; there is no eager fallback.

Not here: TwoBodyDispersionD3 (potentials/dftd3.py) -- its C6 reference table ships as resources/c6.h5 and h5py is
not available in this environment -- and the GELU / bias-free networks of the published ANI-2xr / ANI-2dr parameters
(arch.py:1007-1010; the network kernels implement CELU, which is what ANI-1x / 1ccx / 2x use).
"""
from __future__ import annotations

import ctypes as C  # noqa: F401
import math
import typing as tp

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


class RepulsionXTB(torch.nn.Module):
    """xTB repulsion pair potential (potentials/xtb.py:17-77).  ``krep_hydrogen`` applies to H-H pairs only."""

    def __init__(self, symbols: tp.Sequence[str], krep_hydrogen: float = 1.0, krep: float = 1.5,
                 alpha: tp.Sequence[float] = (), yeff: tp.Sequence[float] = (), *, cutoff: float = math.inf,
                 cutoff_fn: str = "smooth") -> None:
        super().__init__()
        if cutoff_fn not in _lib.CUTOFF_KINDS:
            raise ValueError(f"Unsupported cutoff function {cutoff_fn!r}: the HIP kernels have {sorted(_lib.CUTOFF_KINDS)}")
        if len(symbols) > 7:
            raise ValueError("at most 7 elements (the species field of a neighbor row)")
        self.symbols = tuple(symbols)
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
        self.register_buffer("atomic_numbers", torch.tensor([ATOMIC_NUMBER.get(s, 0) for s in symbols]))
        self.register_buffer("y_ab", torch.outer(y, y))
        self.register_buffer("sqrt_alpha_ab", torch.outer(a, a).sqrt())
        self.register_buffer("k_rep_ab", k)
        self.cutoff = float(cutoff)
        self.cutoff_fn = cutoff_fn
        self._enabled = True
        self._table: tp.Optional[Tensor] = None
        self._own_engine: tp.Optional[AevEngine] = None

    def table(self, device: torch.device) -> Tensor:
        """[8, 8, 4] device table {y_ab, sqrt(alpha_ab), k_ab, 0} (include/anihip.h)."""
        if self._table is None or self._table.device != device:
            S = len(self.symbols)
            t = torch.zeros((8, 8, 4), dtype=torch.float32)
            t[:S, :S, 0], t[:S, :S, 1], t[:S, :S, 2] = self.y_ab.cpu(), self.sqrt_alpha_ab.cpu(), self.k_rep_ab.cpu()
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
        flags = 0 if nbrs.symmetric else _lib.PAIR_PUSH
        _lib.check(_lib.lib().anihip_pair_xtb_repulsion(
            _stream(), species32.numel(), nbrs.lo, nbrs.hi, _ptr(species32), _ptr(nbrs.meta), _ptr(nbrs.ent),
            _ptr(self.table(species32.device)), float(cut), _lib.CUTOFF_KINDS[self.cutoff_fn], flags, _ptr(atomic_e),
            _ptr(grad_coords), _ptr(virial)))

    def compute_from_rows(self, species32: Tensor, coords: Tensor, nbrs: NeighborRows) -> Tensor:
        """Molecular energies [C] (float64), differentiable with respect to coords."""
        return _PairEnergy.apply(coords, self, species32, nbrs)

    def extra_repr(self) -> str:
        return f"symbols={self.symbols}, cutoff={self.cutoff}, cutoff_fn={self.cutoff_fn}"
