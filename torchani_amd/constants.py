"""Model constants of the ANI hot path (values restated from the reference, cited per item).

Reference paths are relative to /root/reference/torchani/.
"""
from __future__ import annotations

import math
import typing as tp

# utils.py:63-66
SYMBOLS_1X: tp.Tuple[str, ...] = ("H", "C", "N", "O")
SYMBOLS_2X: tp.Tuple[str, ...] = ("H", "C", "N", "O", "S", "F", "Cl")
SYMBOLS_2X_ZNUM_ORDER: tp.Tuple[str, ...] = ("H", "C", "N", "O", "F", "S", "Cl")   # utils.py:65 (ANI-2xr / 2dr)
ATOMIC_NUMBER: tp.Dict[str, int] = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9, "S": 16, "Cl": 17}
PADDING_SPECIES = -1  # utils.py:67-74

# constants.py:88-96, ground-state atomic self energies wB97X/6-31G(d), Hartree
# constants.py GSAES["b973c-def2mtzvp"] (ground-state atomic energies at the level of theory of ANI-2dr)
GSAES_B973C_DEF2MTZVP: tp.Dict[str, float] = {
    "H": -0.506930113968, "C": -37.81441001258, "N": -54.556538547322, "O": -75.029181326588,
    "F": -99.688618987039, "S": -398.043159341582, "Cl": -460.082223445159,
}
# resources/atomic_constants.json: (electronegativity, hardness) in eV, for the charge normalizer of ANI-mbis
ELECTRONEGATIVITY_HARDNESS: tp.Dict[str, tp.Tuple[float, float]] = {
    "H": (7.18, 12.84), "C": (6.26, 10.0), "N": (7.27, 14.53), "O": (7.54, 12.16), "F": (10.41, 14.02),
    "S": (6.22, 8.28), "Cl": (8.29, 9.35),
}
# constants.py GSAES["ccsd(t)star-cbs"] (ANI-1ccx) and GSAES["r2scan3c{,_water,_chcl3,_ch3cn}-def2mtzvpp"] (ANI-r2s)
GSAES_CCSDT_STAR_CBS: tp.Dict[str, float] = {
    "H": -0.5, "C": -37.780724507998, "N": -54.515992576387, "O": -74.976148184192,
    "F": -99.624864557142, "S": -397.646401989238, "Cl": -459.664237510771,
}
GSAES_R2SCAN3C: tp.Dict[tp.Optional[str], tp.Dict[str, float]] = {
    None: {"H": -0.49727168567, "C": -37.832225901872, "N": -54.581004402346, "O": -75.057311846055,
           "F": -99.726350798961, "S": -398.08097127572, "Cl": -460.113993263966},
    "water": {"H": -0.494931329259, "C": -37.822388062823, "N": -54.581010824825, "O": -75.059169500763,
              "F": -99.724273365141, "S": -398.082828534447, "Cl": -460.113806300624},
    "chcl3": {"H": -0.496899744403, "C": -37.824548433511, "N": -54.57668102908, "O": -75.056821997619,
              "F": -99.726146486046, "S": -398.085456915563, "Cl": -460.116926115444},
    "ch3cn": {"H": -0.496684906369, "C": -37.824424218755, "N": -54.57657248763, "O": -75.058406925318,
              "F": -99.725926489187, "S": -398.084853327694, "Cl": -460.116392553071},
}
GSAES_WB97X_631GD: tp.Dict[str, float] = {
    "C": -37.8338334,
    "Cl": -460.116700600,
    "F": -99.6949007,
    "H": -0.4993212,
    "N": -54.5732825,
    "O": -75.0424519,
    "S": -398.0814169,
}

# constants.py:78-190 GSAES: ground-state atomic energies (Hartree) by level of theory, the elements the engine supports
GSAES: tp.Dict[str, tp.Dict[str, float]] = {
    "b973c-def2mtzvp": {"H": -0.506930113968, "C": -37.81441001258, "N": -54.556538547322, "O": -75.029181326588, "F": -99.688618987039, "S": -398.043159341582, "Cl": -460.082223445159},
    "wb97x-631gd": {"H": -0.4993212, "C": -37.8338334, "N": -54.5732825, "O": -75.0424519, "F": -99.6949007, "S": -398.0814169, "Cl": -460.1167006},
    "wb97md3bj-def2tzvpp": {"H": -0.498639663159, "C": -37.870597534068, "N": -54.621568655507, "O": -75.111870707635, "F": -99.784869113871, "S": -398.158126819835, "Cl": -460.197921425433},
    "wb97mv-def2tzvpp": {"H": -0.494111111003, "C": -37.844395699666, "N": -54.590952163069, "O": -75.076760965132, "F": -99.745234404775, "S": -398.089446664032, "Cl": -460.124987825603},
    "ccsd(t)star-cbs": {"H": -0.5, "C": -37.780724507998, "N": -54.515992576387, "O": -74.976148184192, "F": -99.624864557142, "S": -397.646401989238, "Cl": -459.664237510771},
    "dsd_blyp_d3bj-def2tzvp": {"H": -0.4990340388250001, "C": -37.812711066967, "F": -99.795668645591, "Cl": -460.052391015914},
    "wb97m_d3bj-def2tzvppd": {"H": -0.4987605100487531, "C": -37.87264507233593, "N": -54.62327513368922, "O": -75.11317840410095, "F": -99.78611622985483, "S": -398.1599636677874, "Cl": -460.1988762285739},
    "revpbe_d3bj-def2tzvp": {"H": -0.504124985686, "C": -37.845615868613, "N": -54.587739850180995, "O": -75.071223222771, "S": -398.041639842051},
    "wb97x-def2tzvpp": {"H": -0.5013925, "C": -37.8459781, "N": -54.5915914, "O": -75.0768759, "F": -99.7471707, "S": -398.1079973, "Cl": -460.1467777},
    "r2scan3c-def2mtzvpp": {"H": -0.49727168567, "C": -37.832225901872, "N": -54.581004402346, "O": -75.057311846055, "F": -99.726350798961, "S": -398.08097127572, "Cl": -460.113993263966},
    "r2scan3c_water-def2mtzvpp": {"H": -0.494931329259, "C": -37.822388062823, "N": -54.581010824825, "O": -75.059169500763, "F": -99.724273365141, "S": -398.082828534447, "Cl": -460.113806300624},
    "r2scan3c_chcl3-def2mtzvpp": {"H": -0.496899744403, "C": -37.824548433511, "N": -54.57668102908, "O": -75.056821997619, "F": -99.726146486046, "S": -398.085456915563, "Cl": -460.116926115444},
    "r2scan3c_ch3cn-def2mtzvpp": {"H": -0.496684906369, "C": -37.824424218755, "N": -54.57657248763, "O": -75.058406925318, "F": -99.725926489187, "S": -398.084853327694, "Cl": -460.116392553071},
}


def linspace(start: float, stop: float, steps: int) -> tp.Tuple[float, ...]:
    """End-point-excluding linspace used for all ANI shifts (utils.py:101-107)."""
    return tuple(start + ((stop - start) / steps) * j for j in range(steps))


class AEVConstants(tp.NamedTuple):
    """Hyper-parameters of the radial/angular symmetry functions."""

    num_species: int
    Rcr: float
    Rca: float
    EtaR: float
    ShfR: tp.Tuple[float, ...]
    EtaA: float
    Zeta: float
    ShfA: tp.Tuple[float, ...]
    ShfZ: tp.Tuple[float, ...]
    cutoff_fn: str = "cosine"  # "cosine" (cutoffs.py:74-81) | "smooth" (CutoffSmooth order 2, cutoffs.py:84-101)

    @property
    def radial_len(self) -> int:
        return self.num_species * len(self.ShfR)

    @property
    def num_species_pairs(self) -> int:
        return self.num_species * (self.num_species + 1) // 2

    @property
    def angular_len(self) -> int:
        return self.num_species_pairs * len(self.ShfA) * len(self.ShfZ)

    @property
    def out_dim(self) -> int:
        return self.radial_len + self.angular_len


def aev_constants_2x(num_species: int = 7, cutoff_fn: str = "cosine") -> AEVConstants:
    # aev/_computer.py:550-600; aev/_terms.py:188-207 (radial), :345-366 (angular)
    return AEVConstants(
        num_species, 5.1, 3.5, 19.7, linspace(0.8, 5.1, 16), 12.5, 14.1, linspace(0.8, 3.5, 8),
        linspace(math.pi / 8, math.pi + math.pi / 8, 4), cutoff_fn,
    )


def aev_constants_simple(num_species: int = 7, cutoff_fn: str = "smooth") -> AEVConstants:
    # arch.py:992-1046 simple_ani defaults (the AEV of ANI-2xr / ANI-2dr): both term families cover_linearly from 0.9 A,
    # radial cutoff 5.2 A, smooth envelope
    return AEVConstants(
        num_species, 5.2, 3.5, 19.7, linspace(0.9, 5.2, 16), 12.5, 14.1, linspace(0.9, 3.5, 8),
        linspace(math.pi / 8, math.pi + math.pi / 8, 4), cutoff_fn,
    )


def aev_constants_1x(num_species: int = 4, cutoff_fn: str = "cosine") -> AEVConstants:
    # aev/_computer.py:498-548
    return AEVConstants(
        num_species, 5.2, 3.5, 16.0, linspace(0.9, 5.2, 16), 8.0, 32.0, linspace(0.9, 3.5, 4),
        linspace(math.pi / 16, math.pi + math.pi / 16, 8), cutoff_fn,
    )


# nn/_containers.py:506-533 (ANI-2x) and :546-570 (ANI-1x): hidden widths per element
HIDDEN_DIMS_2X: tp.Dict[str, tp.Tuple[int, ...]] = {
    "H": (256, 192, 160),
    "C": (224, 192, 160),
    "N": (192, 160, 128),
    "O": (192, 160, 128),
    "S": (160, 128, 96),
    "F": (160, 128, 96),
    "Cl": (160, 128, 96),
}
HIDDEN_DIMS_1X: tp.Dict[str, tp.Tuple[int, ...]] = {
    "H": (160, 128, 96),
    "C": (144, 112, 96),
    "N": (128, 112, 96),
    "O": (128, 112, 96),
}
CELU_ALPHA = 0.1  # nn/_core.py:163-167


# The envelopes the HIP kernels implement, by the reference's names (cutoffs.py:104-121 maps the same strings to its
# classes); anihip_aev_params.cutoff_kind takes the value.
CUTOFF_KERNEL_IDS: tp.Dict[str, int] = {"cosine": 0, "smooth": 1}


def cutoff_kernel_name(cutoff_fn) -> str:
    """``cutoff_fn`` (a name, or any object with a ``_kernel_name`` attribute) -> the name the kernels know."""
    name = cutoff_fn if isinstance(cutoff_fn, str) else getattr(cutoff_fn, "_kernel_name", "")
    if name not in CUTOFF_KERNEL_IDS:
        raise ValueError(f"Unsupported cutoff function {cutoff_fn!r}: the HIP kernels implement {sorted(CUTOFF_KERNEL_IDS)}")
    return name
