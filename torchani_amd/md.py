"""Minimal molecular-dynamics driver around ``ANI.energies_and_forces``.

The production caller of the energy+forces path in the reference is an ASE ``Calculator`` driving ASE's integrators
(torchani/ase.py:32-173, tools/md-benchmark.py); ASE is not available here, so this module carries the two pieces the
benchmark needs: a velocity-Verlet / Langevin (BAOAB) integrator on device tensors and the unit conventions
(Hartree, Angstrom, amu, fs).  Combined with ``neighborlist="verlet_cell_list"`` the pair search is reused between
steps (VerletCellList, neighbors.py:759-884).

Host code only: every step is one stream-ordered ``energies_and_forces`` call plus a handful of elementwise updates.
"""
from __future__ import annotations

import math
import typing as tp

import torch
from torch import Tensor

# CODATA 2018: 1 Ha = 4.3597447222071e-18 J, 1 amu = 1.66053906660e-27 kg  ->  (Ha / Angstrom) / amu in Angstrom / fs^2
ACC_UNIT = 4.3597447222071e-18 / 1e-10 / 1.66053906660e-27 * 1e10 * 1e-30
KB_HARTREE = 3.166811563e-6          # Boltzmann constant, Ha / K
ATOMIC_MASS = {1: 1.008, 6: 12.011, 7: 14.007, 8: 15.999, 9: 18.998, 16: 32.06, 17: 35.45}   # amu


class MolecularDynamics:
    """NVE (velocity Verlet) or NVT (Langevin, BAOAB splitting) dynamics of one system or a batch of molecules.

    species [C, A] (atomic numbers, or element indices if the model was built with periodic_table_index=False --
    then pass ``masses``), coords [C, A, 3] in Angstrom (kept unwrapped), dt in fs.
    """

    def __init__(self, model, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None,
                 pbc: tp.Optional[tp.Sequence[bool]] = None, dt: float = 0.5, masses: tp.Optional[Tensor] = None,
                 temperature: tp.Optional[float] = None, friction: float = 0.002, seed: int = 0) -> None:
        if not coords.is_cuda:
            raise ValueError("MolecularDynamics needs tensors on a ROCm device (no CPU fallback)")
        self.model, self.species, self.cell, self.pbc, self.dt = model, species, cell, pbc, float(dt)
        self.coords = coords.detach().to(torch.float32).clone().contiguous()
        if masses is None:
            if not model.periodic_table_index:
                raise ValueError("pass masses when species are element indices")
            lut = torch.zeros(120, dtype=torch.float32)
            for z, m in ATOMIC_MASS.items():
                lut[z] = m
            masses = lut.to(coords.device)[species.clamp(min=0)]
        self.masses = masses.to(device=coords.device, dtype=torch.float32)
        self.real = (species >= 0)
        self.inv_m = torch.where(self.real, ACC_UNIT / self.masses.clamp(min=1e-6),
                                 torch.zeros_like(self.masses)).unsqueeze(-1)
        self.velocities = torch.zeros_like(self.coords)        # Angstrom / fs
        self.temperature, self.friction = temperature, float(friction)
        self.gen = torch.Generator(device=coords.device).manual_seed(seed)
        self.steps_done = 0
        self._eval()

    def _eval(self) -> None:
        out = self.model.energies_and_forces(self.species, self.coords, self.cell, self.pbc)
        self.potential_energies, self.forces = out.energies, out.forces

    # ---- observables (Hartree, K) ----------------------------------------------------------------------------
    def kinetic_energies(self) -> Tensor:
        ke = 0.5 * (self.masses.unsqueeze(-1) * self.velocities.pow(2)).sum(dim=(1, 2)) / ACC_UNIT
        return ke.double()

    def total_energies(self) -> Tensor:
        return self.potential_energies + self.kinetic_energies()

    def temperatures(self) -> Tensor:
        dof = 3.0 * self.real.sum(dim=1).clamp(min=1).double()
        return 2.0 * self.kinetic_energies() / (dof * KB_HARTREE)

    def set_temperature(self, kelvin: float) -> None:
        """Maxwell-Boltzmann velocities."""
        sigma = torch.sqrt(KB_HARTREE * kelvin * ACC_UNIT / self.masses.clamp(min=1e-6)).unsqueeze(-1)
        noise = torch.randn(self.coords.shape, generator=self.gen, device=self.coords.device)
        self.velocities = torch.where(self.real.unsqueeze(-1), sigma * noise, torch.zeros_like(noise))

    # ---- integrator --------------------------------------------------------------------------------------------
    def run(self, n_steps: int) -> None:
        dt = self.dt
        for _ in range(n_steps):
            self.velocities += (0.5 * dt) * self.forces * self.inv_m
            if self.temperature is None:
                self.coords += dt * self.velocities
            else:   # BAOAB: half drift, Ornstein-Uhlenbeck kick, half drift
                self.coords += (0.5 * dt) * self.velocities
                c1 = math.exp(-self.friction * dt)
                sigma = torch.sqrt(KB_HARTREE * self.temperature * ACC_UNIT * (1.0 - c1 * c1)
                                   / self.masses.clamp(min=1e-6)).unsqueeze(-1)
                noise = torch.randn(self.coords.shape, generator=self.gen, device=self.coords.device)
                self.velocities = torch.where(self.real.unsqueeze(-1), c1 * self.velocities + sigma * noise,
                                              torch.zeros_like(noise))
                self.coords += (0.5 * dt) * self.velocities
            self._eval()
            self.velocities += (0.5 * dt) * self.forces * self.inv_m
            self.steps_done += 1
