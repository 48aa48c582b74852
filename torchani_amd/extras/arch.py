"""Model assembly under the reference's names (torchani/arch.py): ``ANI``, ``ANIq``, the ``Assembler`` and the flexible
builders ``simple_ani`` / ``simple_aniq``.

The Assembler collects the parts step by step like the reference's (arch.py:743-990) and ``assemble(n)`` hands them to the
engine-backed classes of this package; what the kernels do not cover is refused with a ValueError at ``assemble`` time
(the rules are those of ``models.simple_ani``)."""
from __future__ import annotations

import math
import typing as tp

import torch

from ..aev import AEVComputer, ANIAngular, ANIRadial
from ..constants import GSAES, HIDDEN_DIMS_1X, HIDDEN_DIMS_2X, cutoff_kernel_name
from .electro import BaseChargeNormalizer
from ..models import ANI, ANIq, simple_ani, simple_aniq  # noqa: F401
from ..nn import ANINetworks, Ensemble

__all__ = ["ANI", "ANIq", "Assembler", "simple_ani", "simple_aniq"]

_CTORS = {"ani1x": "like_1x", "ani2x": "like_2x", "default": "like_2x"}


def _parse_radial(radial) -> ANIRadial:
    if isinstance(radial, str):
        if radial not in ("ani1x", "ani2x"):
            raise ValueError(f"Unsupported radial term {radial!r} ('ani1x', 'ani2x' or an ANIRadial)")
        return ANIRadial.like_1x() if radial == "ani1x" else ANIRadial.like_2x()
    return radial


def _parse_angular(angular) -> ANIAngular:
    if isinstance(angular, str):
        if angular not in ("ani1x", "ani2x"):
            raise ValueError(f"Unsupported angular term {angular!r} ('ani1x', 'ani2x' or an ANIAngular)")
        return ANIAngular.like_1x() if angular == "ani1x" else ANIAngular.like_2x()
    return angular


def _make_networks(cls, ctor: str, kwargs: tp.Mapping[str, tp.Any], symbols: tp.Sequence[str], in_dim: int):
    """One container from (cls, ctor, kwargs) -- nn/_containers.py:454-544: per-element widths of the recipe."""
    if not (isinstance(cls, type) and issubclass(cls, ANINetworks)):
        raise ValueError(f"the network kernels cover ANINetworks containers, not {cls!r}")
    ctor = _CTORS.get(ctor, ctor)
    if ctor not in ("like_1x", "like_2x"):
        raise ValueError(f"Unsupported network recipe {ctor!r}: 'ani1x' / 'like_1x', 'ani2x' / 'like_2x', 'default'")
    kw = dict(kwargs)
    # (the reference's recipes: CELU with biases for like_1x / like_2x called directly, see nn/_containers.py:479-544)
    activation, bias, out_dim = kw.pop("activation", "celu"), kw.pop("bias", True), kw.pop("out_dim", 1)
    if kw:
        raise ValueError(f"Unsupported network arguments {sorted(kw)}")
    if isinstance(activation, torch.nn.Module):
        activation = {"GELU": "gelu", "CELU": "celu", "TightCELU": "celu"}.get(type(activation).__name__, "")
    if activation not in ("celu", "gelu"):
        raise ValueError("activation 'celu' or 'gelu' (or the corresponding modules)")
    table, other = (HIDDEN_DIMS_1X, (128, 112, 96)) if ctor == "like_1x" else (HIDDEN_DIMS_2X, (160, 128, 96))
    return cls.build(symbols, in_dim, {s: table.get(s, other) for s in symbols}, activation, bias, out_dim=out_dim)


class Assembler:
    """Assembles an ``ANI`` (or ``ANIq``) model step by step (arch.py:743-990)."""

    def __init__(self, symbols: tp.Sequence[str] = (), cls: type = ANI, neighborlist: str = "all_pairs",
                 periodic_table_index: bool = True) -> None:
        self._global_cutoff_fn: str = "smooth"
        self._neighborlist = neighborlist
        self._aev: tp.Optional[tp.Tuple[ANIRadial, ANIAngular, tp.Any, str]] = None
        self._potentials: tp.Dict[str, tp.Tuple[type, tp.Dict[str, tp.Any], float, tp.Any]] = {}
        self._self_energies: tp.Dict[str, float] = {}
        self._container: tp.Optional[tp.Tuple[type, str, tp.Dict[str, tp.Any]]] = None
        self._charge_container: tp.Optional[tp.Tuple[type, str, tp.Dict[str, tp.Any]]] = None
        self._charge_normalizer: tp.Optional[BaseChargeNormalizer] = None
        self._symbols: tp.Tuple[str, ...] = tuple(symbols)
        if not (isinstance(cls, type) and issubclass(cls, ANI)):
            raise ValueError("cls must be ANI or a subclass")
        self._cls = cls
        self.periodic_table_index = periodic_table_index

    # ---- elements and self energies ----
    @property
    def symbols(self) -> tp.Tuple[str, ...]:
        return self._symbols

    def set_symbols(self, symbols: tp.Sequence[str]) -> None:
        self._symbols = tuple(symbols)

    def _check_symbols(self, symbols: tp.Optional[tp.Iterable[str]] = None) -> None:
        if not self.symbols:
            raise ValueError("Please set symbols before setting the gsaes as self energies")
        if symbols is not None and set(self.symbols) != set(symbols):
            raise ValueError(f"Passed symbols don't match supported elements {self._symbols}")

    @property
    def self_energies(self) -> tp.Dict[str, float]:
        if not self._self_energies:
            raise RuntimeError("Self energies have not been set")
        return self._self_energies

    def set_self_energies(self, value: tp.Mapping[str, float]) -> None:
        self._check_symbols(value.keys())
        self._self_energies = dict(value)

    def set_zeros_as_self_energies(self) -> None:
        self._check_symbols()
        self.set_self_energies({s: 0.0 for s in self.symbols})

    def set_gsaes_as_self_energies(self, lot: str = "", functional: str = "", basis_set: str = "") -> None:
        self._check_symbols()
        if (functional and basis_set) and not lot:
            lot = f"{functional}-{basis_set}"
        elif not ((not functional and not basis_set) and lot):
            raise ValueError("Incorrect specification. Either specify *only* lot (preferred) or *both* functional *and* "
                             "basis_set")
        gsaes = GSAES[lot.lower()]
        self.set_self_energies({s: gsaes[s] for s in self.symbols})

    # ---- networks ----
    def set_atomic_networks(self, cls: type = ANINetworks, ctor: str = "ani2x",
                            kwargs: tp.Optional[tp.Dict[str, tp.Any]] = None) -> None:
        self._container = (cls, ctor, dict(kwargs or {}))

    def set_charge_networks(self, cls: type = ANINetworks, ctor: str = "ani2x",
                            kwargs: tp.Optional[tp.Dict[str, tp.Any]] = None,
                            normalizer: tp.Optional[BaseChargeNormalizer] = None) -> None:
        if not issubclass(self._cls, ANIq):
            raise ValueError("Model must be a subclass of ANIq to use charge networks")
        self._charge_normalizer = normalizer
        self._charge_container = (cls, ctor, dict(kwargs or {}))

    # ---- symmetry functions, neighbors, envelopes ----
    def set_aev_computer(self, angular, radial, cutoff_fn="global", strategy: str = "pyaev") -> None:
        radial, angular = _parse_radial(radial), _parse_angular(angular)
        if angular.cutoff > radial.cutoff:
            raise ValueError("Angular cutoff must be smaller or equal to radial cutoff")
        if angular.cutoff <= 0 or radial.cutoff <= 0:
            raise ValueError("Cutoffs must be strictly positive")
        self._aev = (radial, angular, cutoff_fn, strategy)

    def set_neighborlist(self, neighborlist: str) -> None:
        self._neighborlist = neighborlist

    def set_global_cutoff_fn(self, cutoff_fn) -> None:
        self._global_cutoff_fn = cutoff_kernel_name(cutoff_fn)

    def add_potential(self, cls: type, name: str, cutoff: float = math.inf, cutoff_fn="global",
                      kwargs: tp.Optional[tp.Dict[str, tp.Any]] = None) -> None:
        if name in self._potentials or name == "nnp":
            raise ValueError("Potential names must be unique")
        self._potentials[name] = (cls, dict(kwargs or {}), cutoff, cutoff_fn)

    # ---- assembly ----
    def assemble(self, ensemble_size: int = 1, row_capacity: int = 128):
        """Construct the model (arch.py:907-990).  Like the reference's, its networks hold random parameters; CELU networks
        with biases come back trainable, the others frozen (the training passes cover CELU with biases)."""
        if ensemble_size < 1:
            raise ValueError("Ensemble size must be positive")
        if not self.symbols:
            raise RuntimeError("Symbols not set. Call 'set_symbols()' before assembly")
        if self._aev is None:
            raise RuntimeError("AEVComputer not set. Call 'set_aev_computer' before assembly")
        if self._container is None:
            raise RuntimeError("Call 'set_atomic_networks(...)' before assembly")
        radial, angular, cutoff_fn, strategy = self._aev
        envelope = cutoff_kernel_name(self._global_cutoff_fn if cutoff_fn == "global" else cutoff_fn)
        nl = {"all_pairs": "auto", "base": "auto"}.get(self._neighborlist, self._neighborlist)   # (let the engine choose)
        aevc = AEVComputer.from_terms(radial, angular, len(self.symbols), envelope, neighborlist=nl,
                                      row_capacity=row_capacity, strategy=strategy)
        members = [_make_networks(*self._container, self.symbols, aevc.out_dim) for _ in range(ensemble_size)]
        nets = members[0] if ensemble_size == 1 else Ensemble(members)
        saes = [self.self_energies[s] for s in self.symbols]
        if issubclass(self._cls, ANIq):
            if self._charge_container is None:
                raise ValueError("ANIq models need set_charge_networks(...) (merged charge / energy networks are not implemented)")
            qnets = _make_networks(*self._charge_container, self.symbols, aevc.out_dim)
            model = self._cls(self.symbols, aevc, nets, saes, self.periodic_table_index, qnets, self._charge_normalizer)
            qnets.requires_grad_(False)
        else:
            model = self._cls(self.symbols, aevc, nets, saes, self.periodic_table_index)
        for name, (pcls, kw, cutoff, pcut) in self._potentials.items():
            ctor = pcls.from_functional if hasattr(pcls, "from_functional") and "functional" in kw else pcls
            model.add_pair_potential(name, ctor(symbols=self.symbols, **kw, cutoff=cutoff,
                                                cutoff_fn=cutoff_kernel_name(self._global_cutoff_fn if pcut == "global" else pcut)))
        if any(getattr(m.atomics[s], "activation_name", "celu") != "celu" or not m.atomics[s].has_biases
               for m in members for s in self.symbols):
            model.requires_grad_(False)
        return model
