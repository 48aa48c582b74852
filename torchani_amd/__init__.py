"""torchani_amd: MI355X-native engine for the ANI energy+forces hot path, behind torchani's API.

Public names follow the reference (torchani/__init__.py): AEVComputer, ANINetworks (= legacy ANIModel),
Ensemble, SpeciesConverter, SpeciesEnergies, models.ANI1x / models.ANI2x, grad.energies_and_forces.
"""
from . import constants, weights  # noqa: F401  (import-light; no torch needed)

__all__ = ["AEVComputer", "ANINetworks", "ANIModel", "Ensemble", "SpeciesConverter", "SpeciesEnergies",
           "SpeciesAEV", "AtomicNetwork", "models", "grad", "parallel", "utils", "potentials", "ase", "md", "extras",
           "single_point", "SelfEnergy"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so that tooling (fixture generation, the oracle
    # tests) can use torchani_amd.constants / .weights without importing torch
    import importlib

    if name in ("models", "grad", "parallel", "engine", "aev", "nn", "tuples", "_lib", "utils", "potentials", "ase", "md",
                "ops", "extras"):
        return importlib.import_module(f".{name}", __name__)
    if name in ("arch", "io", "electro"):   # host-side conveniences outside the hot path
        return importlib.import_module(f".extras.{name}", __name__)
    table = {
        "AEVComputer": ("aev", "AEVComputer"),
        "ANINetworks": ("nn", "ANINetworks"),
        "ANIModel": ("nn", "ANINetworks"),  # legacy name (nn/_internal.py:13-19)
        "Ensemble": ("nn", "Ensemble"),
        "SpeciesConverter": ("nn", "SpeciesConverter"),
        "AtomicNetwork": ("nn", "AtomicNetwork"),
        "SpeciesEnergies": ("tuples", "SpeciesEnergies"),
        "SpeciesAEV": ("tuples", "SpeciesAEV"),
        "single_point": ("grad", "single_point"),   # (torchani/__init__.py exports these at the top level)
        "SelfEnergy": ("nn", "SelfEnergy"),
    }
    if name in table:
        mod, attr = table[name]
        return getattr(importlib.import_module(f".{mod}", __name__), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
