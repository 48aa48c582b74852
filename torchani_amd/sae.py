"""``torchani.sae`` under its name: the self-energy (per-element energy shift) module lives in torchani_amd.nn."""
from .nn import SelfEnergy  # noqa: F401

__all__ = ["SelfEnergy"]
