"""ANI model assembly with the reference's API surface, plus the fused engine path.

Mirrors torchani/arch.py:302-349 (ANI.forward), :116-126,263-275 (sub-module names), torchani/models.py:
112-119,185-193 (ANI1x / ANI2x recipes) and torchani/grad.py:263-290 (energies_and_forces).

Two ways to evaluate:
  * ``model((species, coords), cell, pbc)`` -> SpeciesEnergies, differentiable through
    torch.autograd (custom Functions wrapping the HIP forward/backward kernels), like the reference;
  * ``model.energies_and_forces(species, coords, cell, pbc)`` -> the fused path used for MD and by
    bench.py: neighbor rows -> AEV -> ensemble fwd+bwd -> AEV backward -> fp64 energy reduction, no
    autograd graph, no host synchronisation, optional sharding of the central atoms over the ranks of a
    torch.distributed (RCCL) process group.
"""
from __future__ import annotations

import math
import typing as tp
import warnings

import numpy as np
import torch
from torch import Tensor

from .aev import AEVComputer
from ._lib import MAX_RAD
from ._lib import MAX_RAD as _lib_MAX_RAD
from .constants import GSAES_WB97X_631GD  # noqa: F401
from . import _lib
from .engine import FIXED_SCALE, _n_cus as _n_cus_of, energy_forces_finish, energy_reduce, fixed_to_float
from .extras.electro import BaseChargeNormalizer, ChargeNormalizer  # noqa: F401
from .nn import ANINetworks, AtomicNetwork, Ensemble, SelfEnergy, SpeciesConverter
from .parallel import join_exact, shard_range, split_exact
from .tuples import EnergiesScalars, AtomicStdev, ForceMagnitudes, ForceStdev, SpeciesEnergiesQBC, SpeciesForces, FusedEnergiesForces, SpeciesEnergies
from .weights import arch_gsaes, arch_networks, arch_spec, random_state_dict


class NNPotential(torch.nn.Module):
    """aev_computer + neural_networks pair (potentials/nnp.py:20-32)."""

    def __init__(self, aev_computer: AEVComputer, neural_networks: torch.nn.Module) -> None:
        super().__init__()
        self.aev_computer = aev_computer
        self.neural_networks = neural_networks
        self._enabled = True


class ANI(torch.nn.Module):
    """ANI-style neural network interatomic potential (arch.py:298-349)."""

    def __init__(self, symbols: tp.Sequence[str], aev_computer: AEVComputer, neural_networks: torch.nn.Module,
                 self_energies: tp.Sequence[float], periodic_table_index: bool = True) -> None:
        super().__init__()
        self.symbols = tuple(symbols)
        self.periodic_table_index = periodic_table_index
        self.species_converter = SpeciesConverter(symbols)
        self.potentials = torch.nn.ModuleDict({"nnp": NNPotential(aev_computer, neural_networks)})
        self.energy_shifter = SelfEnergy(symbols, self_energies)
        self.register_buffer("atomic_numbers", self.species_converter.atomic_numbers.clone())
        self.cutoff = aev_computer.radial.cutoff
        # atoms per launch group of the network stage; None = PackedNetworks.forward_backward's rule: ONE call when its
        # scratch is small (from 65536 atoms on the fused kernel keeps everything but ~100 B per atom in LDS), else groups of
        # 2^20 atoms (8.5 KB of d E/d act0 per atom of a group; a dozen small launches per group)
        self.mlp_chunk: tp.Optional[int] = None
        self.last_collective: tp.Optional[dict] = None   # what the last sharded energies_and_forces all-reduced
        # True: energies_and_forces accumulates forces in int64 fixed point (2^-32 Ha/A): bit-identical results from
        # run to run and for any number of ranks' reduction order, at the price of 24 instead of 12 bytes per atom of
        # accumulator traffic (the reference's cuAEV backward is not reproducible: float atomics, csrc/aev.cu:700-704)
        self.deterministic_forces = False
        # energies_and_forces replays systems of at most this many atoms as a HIP graph once the same species tensor
        # has been seen three times (launch-bound sizes; 0 disables).  From 24 000 atoms on the eager step is the faster one
        # on an MI355X host: it keeps the AEV rows between steps and rewrites only the flagged slabs, which a captured step
        # cannot (46 357-atom solvated protein: 0.94 ms eager, 0.98 replayed; 16 649 atoms: 0.80 either way; round 6)
        self.auto_graph_atoms = 24000
        self._graphs: tp.Dict[tuple, list] = {}

    # arch.py:263-275 convenience accessors
    @property
    def aev_computer(self) -> AEVComputer:
        return self.potentials["nnp"].aev_computer

    @property
    def neural_networks(self):
        return self.potentials["nnp"].neural_networks

    def set_enabled(self, key: str, val: bool = True) -> None:
        # arch.py:136-142
        if key == "energy_shifter":
            self.energy_shifter._enabled = val
        else:
            self.potentials[key]._enabled = val

    def set_strategy(self, strategy: str) -> None:
        """arch.py:145-148: forwarded to the AEV computer (every reference strategy name means the HIP engine here)."""
        self.aev_computer.set_strategy(strategy)

    def to_infer_model(self, use_mnp: bool = False) -> "ANI":
        """arch.py:208-217: the reference swaps its networks for a batched-matmul / C++ inference container; the
        containers here always run the fused native kernels, so this returns the model as it is."""
        self.potentials["nnp"].neural_networks = self.neural_networks.to_infer_model(use_mnp=use_mnp)
        return self

    def _elem_idxs(self, species: Tensor) -> Tensor:
        return self.species_converter(species, nop=not self.periodic_table_index)

    def forward(self, species_coordinates: tp.Tuple[Tensor, Tensor], cell: tp.Optional[Tensor] = None,
                pbc: tp.Optional[Tensor] = None, charge: int = 0, atomic: bool = False,
                ensemble_values: bool = False) -> SpeciesEnergies:
        species, coords = species_coordinates
        if species.dim() != 2 or coords.shape != (species.shape[0], species.shape[1], 3):
            raise ValueError("expected species [C, A] and coords [C, A, 3]")
        assert charge == 0, "Model only supports neutral molecules"
        elem_idxs = self._elem_idxs(species)
        energies = coords.new_zeros(elem_idxs.shape if atomic else elem_idxs.shape[:1])
        if ensemble_values:
            energies = energies.unsqueeze(0)
        extra = None
        if self.potentials["nnp"]._enabled:
            aevs = self.aev_computer(elem_idxs, coords, cell, pbc)
            energies = energies + self.neural_networks(elem_idxs, aevs, atomic, ensemble_values)
            extra = self._scalars_from_aevs(elem_idxs, aevs, charge)
        for name, pot in self.potentials.items():   # pair potentials on the same neighbor rows (arch.py:329-346)
            if name == "nnp" or not pot._enabled:
                continue
            species32 = elem_idxs.to(torch.int32).contiguous()
            nbrs = self.aev_computer.last_neighbors() if self.potentials["nnp"]._enabled else None
            if nbrs is None or pot.cutoff > self.aev_computer.radial.cutoff + 1e-6 or getattr(pot, "needs_all_rows", False):
                nbrs = self._pair_rows(pot, species32, coords, cell, pbc)
            if atomic:
                # per-atom halves of the pair energies (core.py:195-198).  The kernels return the gradient of the SUM
                # only, so this output does not carry one
                if coords.requires_grad and torch.is_grad_enabled():
                    raise NotImplementedError("atomic=True with pair potentials is not differentiable here: "
                                              "use energies_and_forces, or evaluate under torch.no_grad()")
                a = torch.zeros(species32.numel(), dtype=torch.float32, device=coords.device)
                pot.accumulate(species32, nbrs, a, None)
                e_pair = a.view(species32.shape)
            else:
                e_pair = pot.compute_from_rows(species32, coords, nbrs)
            energies = energies + (e_pair.unsqueeze(0) if ensemble_values else e_pair).to(energies.dtype)
        if self.aev_computer.check_overflow and not torch.cuda.is_current_stream_capturing():
            self._raise_on_pair_overflow()
        if self.energy_shifter._enabled:
            energies = energies + self.energy_shifter(elem_idxs, atomic=atomic)
        return self._output(elem_idxs, energies, extra)

    def _scalars_from_aevs(self, elem_idxs: Tensor, aevs: Tensor, charge: int) -> tp.Optional[Tensor]:
        return None   # (ANIq: atomic charges from the same AEVs)

    def _output(self, elem_idxs: Tensor, energies: Tensor, extra: tp.Optional[Tensor]):
        return SpeciesEnergies(elem_idxs, energies)

    def add_pair_potential(self, name: str, pot: torch.nn.Module) -> "ANI":
        """Add a pair potential (torchani_amd.potentials.RepulsionXTB) under ``potentials[name]`` (Assembler.add_potential,
        arch.py:870-893); its energies and forces are included in forward and energies_and_forces."""
        self.potentials[name] = pot
        return self

    def _pair_rows(self, pot, species32: Tensor, coords: Tensor, cell, pbc, lo: int = 0, hi: tp.Optional[int] = None):
        """Neighbor rows for a pair potential whose cutoff exceeds the AEV's radial cutoff (or is infinite): a second
        engine with that cutoff (rows hold at most 256 neighbors per atom)."""
        from .engine import AevEngine

        rc = min(pot.cutoff, 1.0e3)
        if pot._own_engine is None or abs(pot._own_engine.consts.Rcr - rc) > 1e-9:
            pot._own_engine = AevEngine(self.aev_computer.engine().consts._replace(Rcr=rc, Rca=1e-3))
        pbc_t = None if pbc is None else tuple(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc))
        c32 = coords.detach().to(torch.float32).contiguous()
        rows = pot._own_engine.neighbors(species32, c32, cell, pbc_t, lo=lo, hi=hi, mode=self.aev_computer.neighbor_mode,
                                         row_cap=_lib_MAX_RAD)
        pot._last_rows = rows   # (checked with the AEV's rows: an overflowed row was zeroed by the builder)
        return rows

    def _overflow_impossible(self, species: Tensor, cell, pbc) -> bool:
        """Can a neighbor row of this call overflow at all?  Without periodic images an atom's neighbors are atoms of its own
        molecule: with A - 1 <= min(row capacity, 128 angular slots) -- every batch of small molecules -- no row of the network's
        builder or of a pair potential (256 slots) can, and the status word need not be read (the default
        ``check_overflow=True`` then costs no host synchronisation: BASELINE config 2 through the default API)."""
        if species.dim() != 2:
            return False
        if cell is not None and pbc is not None and any(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc)):
            return False
        aevc = self.aev_computer
        if aevc.neighbor_mode not in ("batch", "auto"):
            return False
        return species.shape[1] - 1 <= min(int(aevc.row_capacity), _lib.MAX_ANG)

    def _pair_rows_overflowed(self) -> bool:
        return any(getattr(pot, "_last_rows", None) is not None and pot._last_rows.overflowed()
                   for k, pot in self.potentials.items() if k != "nnp" and pot._enabled)

    def _raise_on_pair_overflow(self) -> None:
        for k, pot in self.potentials.items():
            if k != "nnp" and pot._enabled and getattr(pot, "_last_rows", None) is not None and pot._last_rows.overflowed():
                raise RuntimeError(f"pair potential {k!r}: an atom has more than {_lib_MAX_RAD} neighbors inside its cutoff "
                                   f"({pot.cutoff} A; rows hold at most {_lib_MAX_RAD}): its energies would be wrong")

    # ---- fused path ---------------------------------------------------------------------------------
    @torch.no_grad()
    def energies_and_forces(self, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None,
                            pbc: tp.Optional[tp.Sequence[bool]] = None, group=None,
                            reduce_forces: bool = True, check_overflow: bool = True,
                            shard: tp.Optional[tp.Tuple[int, int]] = None, stress: bool = False) -> FusedEnergiesForces:
        """Energies [C] (float64, NN + self energies) and forces [C, A, 3] without autograd.

        check_overflow (default): read the neighbor builder's status word afterwards (one host sync) and raise --
        after one automatic retry with row_capacity 256 -- if an atom had more neighbors than a row holds; pass
        False inside latency-critical loops / graph capture and call ``aev_computer.last_neighbors().raise_on_overflow()``.

        With a torch.distributed ``group`` (one process per GPU, RCCL) the central atoms are sharded
        contiguously over the ranks; every rank sees all coordinates, evaluates its shard, and the
        partial energies (and, for a shared system, partial forces) are all-reduced.
        ``shard=(rank, world)`` evaluates that shard alone, without any collective (partial energies and
        forces that add up to the full result over the shards).
        ``stress=True`` also returns the virial [3,3] (float64, Hartree) = dE/d strain of all atoms together, summed
        over shards like the energy -- the reference's "fdotr" stress times the volume (ase.py:164-168).
        """
        if not coords.is_cuda:
            raise ValueError("torchani_amd's engine needs tensors on a ROCm device (no CPU fallback)")
        if check_overflow and self._overflow_impossible(species, cell, pbc):
            check_overflow = False   # (nothing to read: no host synchronisation for batches of small molecules)
        out = self._auto_graph_call(species, coords, cell, pbc, group, shard, stress, check_overflow)
        if out is not None:
            return out
        elem_idxs = self._elem_idxs(species)
        species32 = elem_idxs.to(torch.int32).contiguous()
        c32 = coords.detach().to(torch.float32).contiguous()
        n_central = species32.numel() if group is None and shard is None else 0   # (shards: the library's default)
        hint = 0 if torch.cuda.is_current_stream_capturing() else self._tile_hint(species, species32, n_central)
        if self.two_product_backward:
            hint |= _lib.MLP_FLAG_BWD_TWO_PRODUCTS
        out = self._energies_and_forces_core(species32, c32, cell, pbc, group, reduce_forces, False, shard, stress, hint,
                                             species)
        if check_overflow and not torch.cuda.is_current_stream_capturing():
            # one host sync after everything is queued: a row over capacity was zeroed by the builder, the result
            # would be silently wrong (the reference asserts on the device, csrc/aev.cu:229).  Retry once at the
            # largest row capacity, then raise.
            aevc = self.aev_computer
            if aevc.last_neighbors().overflowed():
                if aevc.row_capacity < MAX_RAD:
                    warnings.warn(f"neighbor rows overflowed row_capacity={aevc.row_capacity}: retrying with {MAX_RAD}")
                    aevc.row_capacity = MAX_RAD
                    out = self._energies_and_forces_core(species32, c32, cell, pbc, group, reduce_forces, False, shard,
                                                         stress, hint, species)
                aevc.last_neighbors().raise_on_overflow()
            self._raise_on_pair_overflow()
        return out

    def _auto_graph_call(self, species, coords, cell, pbc, group, shard, stress, check_overflow):
        """Small systems are launch-bound (a dozen kernels of a few microseconds): from the third call with the SAME species
        tensor (identity and version -- an MD loop, a batch re-evaluated with new coordinates) and shapes, the step is
        replayed as one HIP graph (GraphedEnergiesForces); results are copied out of the graph's static buffers.
        Returns None when the call does not qualify (large system, sharded, stress, capture in progress, ...)."""
        n = species.numel()
        if (self.auto_graph_atoms <= 0 or n > self.auto_graph_atoms or group is not None or shard is not None or stress
                or self.deterministic_forces or self.aev_computer.verlet is not None   # (its displacement check syncs)
                or torch.cuda.is_current_stream_capturing()):
            return None
        pbc_key = None if pbc is None else tuple(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc))
        key = (species.data_ptr(), species._version, tuple(species.shape), coords.device, cell is None, pbc_key,
               self._config_stamp())
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            # (the entry keeps the species tensor alive: while it exists no other tensor can show up under its address, so
            # an equal key always means the same tensor with the same contents)
            self._graphs[key] = ent = [0, None, species]
        ent[0] += 1
        if ent[0] < 3:
            return None
        if ent[1] is None:
            try:
                ent[1] = GraphedEnergiesForces(self, species, coords, cell, pbc)
            except RuntimeError as err:   # something on the path cannot be captured: stay eager for good
                warnings.warn(f"HIP graph capture of energies_and_forces failed ({err}); continuing without graphs")
                self.auto_graph_atoms = 0
                self._graphs.clear()
                return None
        out = ent[1](coords.detach(), cell)
        res = FusedEnergiesForces(out.energies.clone(), out.forces.clone(), out.atomic_energies.clone(), None)
        if check_overflow:
            nb = self.aev_computer.last_neighbors()
            if nb.overflowed() or self._pair_rows_overflowed():   # (let the eager path retry with larger rows / raise)
                self._graphs.pop(key, None)
                return None
        return res

    def _config_stamp(self) -> tuple:
        """Everything besides the inputs and the parameters that a captured step depends on: a change of any of it must
        not replay an old graph."""
        aevc = self.aev_computer
        pots = tuple((k, id(p), bool(p._enabled), getattr(p, "_version_stamp", 0)) for k, p in self.potentials.items())
        return (pots, bool(self.energy_shifter._enabled), aevc.row_capacity, aevc.neighbor_mode, self.mlp_chunk,
                id(self.neural_networks))

    def _tile_hint(self, species: Tensor, elem_idxs: Tensor, n_central: int) -> int:
        """Layer-0 backward tiling for mid-size systems.  From 16 384 atoms on the library tiles 256 rows; that is right
        for water-like compositions (few AEV slabs per tile: the skinny kernel, faster at every size measured), but
        with four or more elements present the 128-row 8-wave kernel is 13-18 % faster per step at 16 k - 46 k atoms
        (1C17, solvated 1hz5: DESIGN.md section 6), where 256-row tiles are too few to balance over the CUs.  The number of
        elements present costs one host sync per distinct ``species`` tensor: cached by identity and version, with a
        reference to the tensor so that its address cannot be handed to another one meanwhile."""
        if n_central < 16384:
            return 0
        key = (species.data_ptr(), species._version, tuple(species.shape))
        hit = self.__dict__.get("_n_elem_cache")
        if hit is None or hit[0] != key:
            present = torch.bincount(elem_idxs.reshape(-1).clamp(min=-1) + 1, minlength=len(self.symbols) + 1)[1:]
            counts = present.tolist()
            hit = (key, sum(c > 0 for c in counts), species, counts)
            self.__dict__["_n_elem_cache"] = hit
        if n_central < 24000:
            return _lib.MLP_FLAG_SMALL_TILES if hit[1] >= 4 else 0
        # Large systems (layer-0 backward inside the fused kernel): one launch per species with compile-time network widths is
        # 5-6 % faster per tile (round 6), but every launch ends with a partly filled last round of the CUs and a species of a
        # handful of atoms still costs a whole tile through all members; the one-launch kernel hands its tiles out by falling
        # cost instead (mid-size systems).  Both are priced in units of "one hydrogen tile through the ensemble" and the
        # cheaper one is taken: 2.34 M-atom water box 143 rounds in 2 launches (per-species); the 46 k-atom solvated protein
        # 5 launches of 2 + 1 + 1 + 1 + 1 rounds against 3.0 of the queue (one launch); water boxes of 24 k / 41 k / 81 k /
        # 192 k atoms: per-species / per-species / one launch / per-species -- each as measured.
        return _lib.MLP_FLAG_SHAPED if self._per_species_launches_pay(hit[3], _n_cus_of(elem_idxs.device) if elem_idxs.is_cuda else 256) else 0

    def _per_species_launches_pay(self, counts: tp.Sequence[int], n_cus: int) -> bool:
        nets = self.neural_networks
        members = nets._member_networks() if hasattr(nets, "_member_networks") else []
        if not members or not hasattr(members[0], "atomics"):
            return False
        cost, tiles = [], []
        for sym, cnt in zip(self.symbols, counts):
            if cnt <= 0:
                continue
            lins = members[0].atomics[sym].linears()
            if len(lins) != 4:
                return False
            h1, h2, h3 = (lin.out_features for lin in lins[:3])
            # a tile of this species relative to the ANI-2x hydrogen network (256 / 192 / 160 over four flagged slabs): a quarter
            # of an item does not scale with the widths (measured: oxygen 192 / 160 / 128 at 0.75)
            cost.append(0.25 + 0.75 * (128 * h1 + h1 * h2 + h2 * h3) / 112640.0)
            tiles.append((cnt + 63) // 64)
        if not tiles:
            return False
        per_species = sum(-(-t // n_cus) * c for t, c in zip(tiles, cost))
        if (sum(counts) + 63) // 64 + len(self.symbols) >= 4 * n_cus:
            # (from four rounds of tiles on the library draws the tiles of a per-species launch from a queue and lets the next
            # species' launch fill the CUs as they come free: no partly filled rounds, one tile of tail)
            per_species = sum(t * c for t, c in zip(tiles, cost)) / n_cus + 0.6
        # the queue: longest-processing-time-first over the workgroups, class by class
        import numpy as np

        load = np.zeros(n_cus)
        for c, t in sorted(zip(cost, tiles), reverse=True):
            load += (t // n_cus) * c
            r = t % n_cus
            if r:
                load[np.argpartition(load, r - 1)[:r]] += c
        one_launch = 1.055 * float(load.max())   # (run-time network widths)
        return per_species < one_launch

    # The AEV rows of energies_and_forces are internal: they live in buffers the engine keeps between steps and updates in
    # place (AevEngine.forward_update: zeros are written once, a step rewrites only the slabs that were or are flagged --
    # 0.6 KB instead of 4 KB per water atom) -- from the second consecutive call with the same atom count and central range
    # on (AevEngine.rows_wanted: calls whose sizes change every time pin and memset nothing); ``release_aev_rows()`` frees
    # them (4 KB per central atom, 9.4 GB at 2.34 M atoms).  False: a fresh, fully written buffer per call.
    keep_aev_rows = True

    # OFF by default (anihip.h, ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS): the backward GEMMs of the large-system network kernel
    # (>= 65 536 atoms) with two products instead of three -- energies unchanged, forces within ~1e-6 Ha/A of the default's
    # (inside the 1e-4 Ha/A parity gate, outside this package's 5e-6 regression gate), a sixth fewer MFMAs.
    two_product_backward = False

    def release_aev_rows(self) -> None:
        self.aev_computer.engine().release_rows()

    @staticmethod
    def _plain_slabs(eng, packed) -> bool:
        """A general symmetry-function grid (AEVComputer.from_constants) whose networks run through the fused kernel: the
        general AEV kernel flags the plain 32-column slabs of a row that can be non-zero and layer 0 skips the others."""
        return (not eng.tuned) and packed.radial_len == 0 and packed.precision == "f16x3" and eng.L <= 1024 and eng.L % 4 == 0

    # ---- species numbered "present ones first" inside the engine ------------------------------------------------------
    compact_species = True   # large systems: relabel the species so that the AEV blocks of absent species come last

    def _engine_species(self, species32: Tensor, key_tensor: tp.Optional[Tensor] = None
                        ) -> tp.Tuple[Tensor, tp.Optional[tp.Tuple[int, ...]]]:
        """The species indices the kernels work with, and their order (None: as given).

        The layer-0 GEMMs skip the 32-column AEV slabs no atom of a tile has a neighbor for.  Radial blocks are 16 columns
        per species, two species to a slab: water under ANI-2x (H = 0, O = 3 of H C N O S F Cl) touches two half-empty
        radial slabs.  Relabelling the species of a SYSTEM "present ones first, by falling abundance" (H -> 0, O -> 1) puts its radial blocks
        side by side -- 4 flagged slabs instead of 5 -- and costs nothing but a permutation of the first-layer weights
        (nn.ANINetworks._pack(species_order=...)): the AEV rows are internal to energies_and_forces, nobody sees their
        column order.  Only for systems large enough that the one host read of the species histogram (cached per species
        tensor, as for _tile_hint: from 16 384 atoms on) does not matter, and only while the networks are the only consumer of the neighbor rows' species codes.
        key_tensor: the caller's own species tensor (any dtype) the cache entry is tied to, like _tile_hint."""
        n = species32.numel()
        key_tensor = species32 if key_tensor is None else key_tensor
        if (not self.compact_species or n < 16384
                or any(k != "nnp" and p._enabled for k, p in self.potentials.items())):
            return species32, None
        aevc = self.aev_computer
        if (len(aevc.radial.shifts), len(aevc.angular.shifts) * len(aevc.angular.sections)) != (16, 32):
            # (a general symmetry-function grid has no 16 / 32-column blocks to line up with the 32-column slabs, and the
            # packer's column permutation is written for them: nn.ANINetworks._pack(species_order=...) would refuse)
            return species32, None
        if torch.cuda.is_current_stream_capturing():
            # (a capture must own the tensors it records -- GraphedEnergiesForces asks before it captures and passes its
            # own copy down; anything else that captures gets the numbering as given)
            return species32, None
        key = (key_tensor.data_ptr(), key_tensor._version, tuple(key_tensor.shape))
        hit = self.__dict__.get("_species_order_cache")
        if hit is None or hit[0] != key:
            S = self.aev_computer.num_species
            present = torch.bincount(species32.reshape(-1).clamp(min=0), minlength=S)[:S]
            present[0] -= (species32 < 0).sum()   # (padding atoms were counted as species 0)
            counts = present.tolist()
            have = [c > 0 for c in counts]
            # the present species by falling abundance: the two most frequent elements of the system share the first radial
            # slab (a solvated protein H C N O S: the water's H and O, so that its atoms away from the solute flag four slabs,
            # the fused kernel's single-pass case, instead of five)
            order = tuple(sorted((s for s in range(S) if have[s]), key=lambda s: (-counts[s], s))
                          + [s for s in range(S) if not have[s]])
            # worth it only if the frequent species then sit closer together: sum over the radial slabs (two species each) of
            # the count of the slab's most frequent species
            weight = lambda o: sum(max(counts[s] for s in o[k:k + 2]) for k in range(0, S, 2))   # noqa: E731
            if order == tuple(range(S)) or weight(order) >= weight(tuple(range(S))):
                hit = (key, key_tensor, None, None)
            else:
                lut = torch.full((S + 1,), -1, dtype=torch.int32, device=species32.device)
                lut[1 + torch.tensor(order, device=species32.device)] = torch.arange(S, dtype=torch.int32, device=species32.device)
                hit = (key, key_tensor, order, lut[(species32 + 1).long()].contiguous())
            self.__dict__["_species_order_cache"] = hit
        return (species32 if hit[2] is None else hit[3]), hit[2]

    def _energies_and_forces_core(self, species32: Tensor, c32: Tensor, cell, pbc, group, reduce_forces,
                                  check_overflow, shard, stress: bool = False, tile_hint: int = 0,
                                  species_key: tp.Optional[Tensor] = None,
                                  engine_species: tp.Optional[tp.Tuple[Tensor, tp.Optional[tp.Tuple[int, ...]]]] = None
                                  ) -> FusedEnergiesForces:
        """The stream-ordered part of energies_and_forces (element indices int32, coords fp32 contiguous):
        no host synchronisation unless check_overflow, so it can be captured into a HIP graph."""
        C, A = species32.shape
        n = C * A
        if (group is None and shard is None and C == 1 and n >= 65536 and self.locality_sort != "never"
                and not torch.cuda.is_current_stream_capturing() and self._spatial_ok(C, n)
                and self._wants_locality_sort(species32, c32, cell, pbc, species_key)):
            shard = (0, 1)   # one "rank" owning everything: the spatial path works on the cell-sorted copy
        if (group is not None or shard is not None) and self._spatial_ok(C, n):
            return self._energies_and_forces_spatial(species32, c32, cell, pbc, group, reduce_forces, check_overflow, shard,
                                                     stress, tile_hint, species_key)
        lo, hi = shard_range(n, group) if shard is None else shard_range(n, rank=shard[0], world=shard[1])
        aevc = self.aev_computer
        eng = aevc.engine()
        pbc_t = None if pbc is None else tuple(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc))
        given = species32
        # (the kernels' numbering of the species; `given` indexes the self energies)
        species32, order = engine_species if engine_species is not None else self._engine_species(given, species_key)
        nbrs = aevc.neighbor_rows(species32, c32, cell, pbc_t, lo=lo, hi=hi)
        packed = self.neural_networks._pack(c32.device, order)
        # per-atom flags of the AEV slabs that are not identically zero (absent neighbor species): the
        # layer-0 GEMMs skip the others
        slab_mask = None
        # AEV rows and their gradients exist for this rank's central atoms only ([hi - lo, L] buffers)
        plain = self._plain_slabs(eng, packed)
        if packed.radial_len == 16 * eng.params.num_species and eng.tuned and eng.n_slabs <= 32:
            if self.keep_aev_rows and not torch.cuda.is_current_stream_capturing() and eng.rows_wanted(n, lo, hi, c32.device):
                # rows and flags in the engine's kept buffers, updated in place (AevEngine.forward_update): from the second
                # consecutive call with these sizes on
                aev, slab_mask = eng.forward_update(species32, nbrs)
            else:
                # (the AEV kernel writes the flags of every central atom; the others are read by nobody, zero for tidiness)
                slab_mask = (torch.empty if (lo == 0 and hi == n) else torch.zeros)(n, dtype=torch.int32, device=c32.device)
                aev = eng.forward(species32, nbrs, slab_mask=slab_mask, shard_rows=True)
        else:
            if plain:   # a general grid: flags of the plain 32-column slabs from the general AEV kernel
                slab_mask = torch.zeros(n, dtype=torch.int32, device=c32.device)
            aev = eng.forward(species32, nbrs, slab_mask=slab_mask, shard_rows=True)
        atomic_e, grad_aev, _ = packed.forward_backward(species32, aev, lo=lo, hi=hi, want_grad=True,
                                                        chunk=self.mlp_chunk, slab_mask=slab_mask,
                                                        shard_rows=True, tile_hint=tile_hint, plain_slabs=plain)
        if plain:
            slab_mask = None   # (the general AEV backward reads the blocks of present species only: no flags needed)
        virial = torch.empty((3, 3), dtype=torch.float64, device=c32.device) if stress else None
        pair_e, pair_g, pair_w = self._pair_terms(species32, c32, cell, pbc_t, nbrs, lo, hi, stress)
        from .parallel import FORCE_COLLECTIVES

        world = 1 if group is None else torch.distributed.get_world_size(group)
        several = world > 1 or (group is not None and FORCE_COLLECTIVES)
        sae = self._sae64(c32.device) if self.energy_shifter._enabled else None
        if self.deterministic_forces:
            # order-independent sums: int64 fixed-point accumulators (2^-32) for the forces (ANIHIP_BWD_FIXED_POINT), and
            # for a sharded run ONE int64 all-reduce that also carries energies and virial at the same resolution
            n_tail = (C + (9 if stress else 0)) if several else 0
            red = torch.zeros(3 * n + n_tail, dtype=torch.int64, device=c32.device)
            eng.backward(species32, nbrs, grad_aev, grad_coords=red[:3 * n].view(n, 3), shard_rows=True,
                         virial=virial, slab_mask=slab_mask, fixed_point=True)
            if pair_g is not None:   # (computed without atomics: deterministic as well)
                red[:3 * n] += torch.round(pair_g.reshape(-1).to(torch.float64) / FIXED_SCALE).to(torch.int64)
                if stress:
                    virial += pair_w
            energies = energy_reduce(given, atomic_e if pair_e is None else atomic_e + pair_e, sae, lo, hi)
            if several:
                red[3 * n:3 * n + C] = torch.round(energies / FIXED_SCALE).to(torch.int64)
                if stress:
                    red[3 * n + C:] = torch.round(virial.reshape(-1) / FIXED_SCALE).to(torch.int64)
                torch.distributed.all_reduce(red if reduce_forces else red[3 * n:], group=group)
                energies = red[3 * n:3 * n + C].to(torch.float64) * FIXED_SCALE
                if stress:
                    virial = (red[3 * n + C:].to(torch.float64) * FIXED_SCALE).view(3, 3)
                self.last_collective = {"collectives_per_step": 1, "world_size": world,
                                        "bytes": 8 * (red.numel() if reduce_forces else n_tail)}
            forces = fixed_to_float(red[:3 * n]).neg_().view(C, A, 3)
        else:
            # forces are accumulated straight into the buffer that a sharded run all-reduces: [3 n forces | 4 C energy
            # parts | 36 virial parts]
            n_tail = (4 * C + (36 if stress else 0)) if several else 0
            red = torch.zeros(3 * n + n_tail, dtype=torch.float32, device=c32.device)
            grad_coords = eng.backward(species32, nbrs, grad_aev, grad_coords=red[:3 * n].view(n, 3), shard_rows=True,
                                       virial=virial, slab_mask=slab_mask)
            if pair_g is not None:
                grad_coords += pair_g
                if stress:
                    virial += pair_w
            # (energies and the sign flip of the gradient share the last launch of the step)
            energies = energy_forces_finish(given, atomic_e if pair_e is None else atomic_e + pair_e, sae,
                                            grad_coords, lo, hi)
            forces = grad_coords.view(C, A, 3)
            if several:
                # ONE collective per step: the fp64 partial energies (and virial) ride in the fp32 force buffer as four
                # exactly-summable fp32 parts each (parallel.split_exact), so the sum over ranks is exact and
                # independent of the reduction order
                red[3 * n:3 * n + 4 * C] = split_exact(energies).reshape(-1)
                if stress:
                    red[3 * n + 4 * C:] = split_exact(virial.reshape(-1)).reshape(-1)
                torch.distributed.all_reduce(red if reduce_forces else red[3 * n:], group=group)
                energies = join_exact(red[3 * n:3 * n + 4 * C].view(C, 4))
                if stress:
                    virial = join_exact(red[3 * n + 4 * C:].view(9, 4)).view(3, 3)
                self.last_collective = {"collectives_per_step": 1, "world_size": world,
                                        "bytes": 4 * (red.numel() if reduce_forces else n_tail)}
        if check_overflow:
            nbrs.raise_on_overflow()
        aevc._last_neighbors = nbrs
        return FusedEnergiesForces(energies, forces, atomic_e.view(C, A), virial)

    # ---- one big system on several ranks: spatial shards + halo (parallel.SpatialShards) ---------------------------
    partition = "spatial"   # "index": contiguous index ranges + one all-reduce of the whole force array (round-2 scheme)
    partition_skin = 0.0    # > 0 (Angstrom): halos that much wider, the partition is kept until an atom has moved skin / 2

    # The AEV backward gathers 64 B of every neighbor's gradient row: with atoms in a spatially coherent order (an MD
    # engine's, a lattice's) those rows are in cache, with a shuffled order they are not -- 7.8 instead of 4.7 ms at 2.34 M
    # atoms (tools/kbench.py --order shuffle).  "auto": a large single system whose order is NOT coherent is evaluated on
    # a cell-sorted copy (the spatial-shard machinery with one rank: gather, kernels, scatter back; the sorted order is
    # kept until an atom has moved half a cell).  "always" / "never" force the choice.
    locality_sort = "auto"

    def _wants_locality_sort(self, species32: Tensor, c32: Tensor, cell, pbc, key_tensor: tp.Optional[Tensor]) -> bool:
        if self.locality_sort == "always":
            return True
        key_tensor = species32 if key_tensor is None else key_tensor
        key = (key_tensor.data_ptr(), key_tensor._version, tuple(key_tensor.shape))
        hit = self.__dict__.get("_locality_cache")
        if hit is None or hit[0] != key:
            pbc_t = None if pbc is None else tuple(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc))
            part = self._spatial_partition(species32.view(-1), c32, cell, pbc_t, 0, 1, key_tensor)
            # neighbors in the cell-sorted order that are also near each other in the given order
            # (a random order puts 2 / 256 of them within n / 256 places; a lattice or an MD engine's order most of them)
            near = ((part.order[1:] - part.order[:-1]).abs() < max(64, species32.numel() // 256)).float().mean()
            hit = (key, key_tensor, bool(float(near) < 0.05))
            self.__dict__["_locality_cache"] = hit
        return hit[2]

    def _spatial_reach(self) -> float:
        """How far beyond its slab a rank must see for every enabled potential (Angstrom; inf: cannot be cut into slabs).

        Networks: the AEV's radial cutoff.  An analytic pair potential on rows of the owned atoms: its own cutoff (what it
        pushes onto halo atoms travels with the halo force rows).  D3 (``needs_all_rows``): THREE cutoffs -- its kernel
        gathers the complete gradient on an owned atom k from k's neighbors j (one cutoff), whose d E / d CN_j sums over
        j's own neighbors m (two), whose coordination numbers CN_m count m's neighbors (three); with all of that inside
        the local system the owned atoms' D3 energies and forces are exact and need nothing from another rank."""
        reach = self.aev_computer.radial.cutoff
        for k, p in self.potentials.items():
            if k == "nnp" or not p._enabled:
                continue
            reach = max(reach, (3.0 if getattr(p, "needs_all_rows", False) else 1.0) * float(p.cutoff))
        return reach

    # How a kept partition (partition_skin > 0, more than one rank) is verified.  "lagged": every step queues the validity
    # flags of its coordinates on the device and reads those of the PREVIOUS step -- no host synchronisation in a steady MD
    # loop; a step is renewed one step after an atom has moved 0.8 x skin / 2, and a jump of more than the rest of the skin
    # inside ONE step (frame replay, Monte-Carlo moves, optimizer jumps, a changed cell) is only noticed by the NEXT call,
    # which raises -- call ``check_partition()`` behind the last step of such a loop.  The first moved step after a cut is
    # always verified at once.  "strict": every step reads its own flags before it is evaluated (one host synchronisation
    # per step) and renews the partition in the same step: never a stale halo, for drivers that move atoms arbitrarily.
    partition_check = "lagged"

    def check_partition(self) -> None:
        """Raise if the LAST spatially sharded step was evaluated on a partition its coordinates had outrun (the flags that
        step queued are otherwise read by the next call; ``partition_check = "lagged"``)."""
        hit = self.__dict__.get("_spatial_cache")
        if hit is None:
            return
        _, invalid = hit[1].poll()
        if invalid and hit[1].world > 1:
            self.__dict__["_spatial_cache"] = None
            raise RuntimeError("spatial shards: an atom moved more than partition_skin / 2 = "
                               f"{0.5 * self.partition_skin:.3f} A before the partition was renewed -- the last step was "
                               "evaluated with too narrow a halo. Use a larger partition_skin, a smaller step, or "
                               "partition_check = 'strict'.")

    def _spatial_ok(self, C: int, n: int) -> bool:
        """Slab decomposition applies to ONE system evaluated through its own pair search, with every enabled potential
        of finite range (the halo is as wide as the widest of them needs, _spatial_reach; a cutoff-free potential falls
        back to index ranges + one all-reduce of the whole force array)."""
        if self.partition != "spatial" or C != 1 or n < 2 or self.aev_computer.verlet is not None:
            return False
        return math.isfinite(self._spatial_reach())

    def _spatial_partition(self, species32: Tensor, c32: Tensor, cell, pbc_t, rank: int, world: int,
                           species_key: tp.Optional[Tensor] = None):
        """This rank's SpatialShards for the given coordinates: cut once and kept while it is valid.

        Same coordinate tensor (identity and version), same species, same box: the cached partition, no device work.  Moved
        coordinates with ``partition_skin`` > 0: the partition is kept until an atom has moved 0.8 x skin / 2 since it was
        cut -- decided WITHOUT a host synchronisation: every step queues the validity flags for its coordinates
        (SpatialShards.check_async) and reads the flags of the PREVIOUS step (poll), like the overflow word of the neighbor
        rows.  Every rank sees the same coordinates and takes the same decision.  Should an atom have outrun the whole skin
        within that one step of lag, the late read raises (the step before it was evaluated with too narrow a halo)."""
        from .parallel import SpatialShards

        spk = species32 if species_key is None else species_key
        # (the species decide which atoms are padding: those are sorted last and kept out of every halo, so a partition cut
        # for one species tensor must not serve another)
        key = (c32.data_ptr(), c32._version, tuple(c32.shape), None if cell is None else (cell.data_ptr(), cell._version),
               pbc_t, rank, world, spk.data_ptr(), spk._version, tuple(spk.shape))
        hit = self.__dict__.get("_spatial_cache")
        if hit is not None and hit[0] != key and (self.partition_skin > 0.0 or world == 1) and hit[0][2] == key[2] and \
                hit[0][4:] == key[4:] and (cell is None) == (hit[3] is None):
            part = hit[1]
            if world > 1 and (self.partition_check == "strict" or not part.check_pending):
                # same-step guard: nothing is known yet about coordinates that moved since the cut (the first moved step
                # after it; a caller that is not in a steady loop), or the caller wants every step verified before it is
                # evaluated -- read the flags of THESE coordinates now (one host synchronisation) and renew at once
                part.poll()
                renew, invalid = part.check_now(c32, cell)[0], False
            else:
                renew, invalid = part.poll()   # (what the previous step found out about ITS coordinates)
            if invalid and world > 1:
                self.__dict__["_spatial_cache"] = None
                raise RuntimeError(
                    "spatial shards: an atom moved more than partition_skin / 2 = "
                    f"{0.5 * self.partition_skin:.3f} A before the partition was renewed -- the previous step was evaluated "
                    "with too narrow a halo. Use a larger partition_skin (or a smaller time step).")
            if not renew:
                # moved coordinates, same box, nobody further than 0.8 x skin / 2 from where the partition was cut (as of the
                # previous step): keep it, and queue the same question for these coordinates
                part.check_async(c32, cell)
                hit = (key, part, c32, cell, spk)
                self.__dict__["_spatial_cache"] = hit
        if hit is None or hit[0] != key:
            # (the entry keeps the tensors alive, so an equal key means the same coordinates, not a recycled address)
            # (one rank has no halo: any order is correct, the skin only says when the order has stopped being local)
            skin = self.partition_skin if world > 1 else max(self.partition_skin, self.aev_computer.radial.cutoff)
            hit = (key, SpatialShards(c32, cell, pbc_t, world, rank, self._spatial_reach(), species32, skin=skin),
                   c32, cell, spk)
            self.__dict__["_spatial_cache"] = hit
        return hit[1]

    def _energies_and_forces_spatial(self, species32: Tensor, c32: Tensor, cell, pbc, group, reduce_forces, check_overflow,
                                     shard, stress: bool, tile_hint: int, species_key: tp.Optional[Tensor] = None
                                     ) -> FusedEnergiesForces:
        """energies_and_forces of ONE system sharded spatially: this rank evaluates the central atoms of its slab on the
        local system [left halo | owned | right halo], one all-gather of the halo force rows (+ partial energy / virial)
        completes its owned atoms' forces.  reduce_forces=True additionally gathers every rank's owned forces and
        per-atom energies (input order); False leaves them distributed (rows of other ranks are zero)."""
        n = species32.numel()
        if shard is None:
            rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
        else:
            rank, world = shard
        pbc_t = None if pbc is None else tuple(bool(b) for b in (pbc.tolist() if isinstance(pbc, Tensor) else pbc))
        part = self._spatial_partition(species32, c32, cell, pbc_t, rank, world, species_key)
        sp_e, order = self._engine_species(species32, species_key)   # (sp_given indexes the self energies)
        sp_given = part.local(species32).view(1, -1).contiguous()
        sp_l = sp_given if order is None else part.local(sp_e).view(1, -1).contiguous()
        x_l = part.local(c32, 3).view(1, -1, 3).contiguous()
        nl = part.n_local
        lo, hi = part.n_left, part.n_left + part.n_owned
        aevc = self.aev_computer
        eng = aevc.engine()
        dev = c32.device
        nbrs = aevc.neighbor_rows(sp_l, x_l, cell, pbc_t, lo=lo, hi=hi)
        packed = self.neural_networks._pack(dev, order)
        slab_mask = None
        plain = self._plain_slabs(eng, packed)
        if packed.radial_len == 16 * eng.params.num_species and eng.tuned and eng.n_slabs <= 32:
            if self.keep_aev_rows and not torch.cuda.is_current_stream_capturing() and eng.rows_wanted(nl, lo, hi, dev):
                aev, slab_mask = eng.forward_update(sp_l, nbrs)   # (kept buffers, updated in place)
            else:
                slab_mask = torch.zeros(nl, dtype=torch.int32, device=dev)
                aev = eng.forward(sp_l, nbrs, slab_mask=slab_mask, shard_rows=True)
        else:
            if plain:
                slab_mask = torch.zeros(nl, dtype=torch.int32, device=dev)
            aev = eng.forward(sp_l, nbrs, slab_mask=slab_mask, shard_rows=True)
        if world > 1 and not torch.cuda.is_current_stream_capturing() and part.n_owned >= 24000:
            # a rank's own launch scheme of the network stage (_tile_hint prices whole systems): from the composition of the atoms
            # it owns, worked out once per partition (one host read: the partition is cut once per skin of motion)
            hint_l = getattr(part, "_tile_hint_owned", None)
            if hint_l is None:
                counts = torch.bincount(sp_given.view(-1)[lo:hi].clamp(min=-1) + 1, minlength=len(self.symbols) + 1)[1:].tolist()
                hint_l = _lib.MLP_FLAG_SHAPED if self._per_species_launches_pay(counts, _n_cus_of(dev)) else 0
                part._tile_hint_owned = hint_l
            tile_hint |= hint_l
        atomic_e, grad_aev, _ = packed.forward_backward(sp_l, aev, lo=lo, hi=hi, want_grad=True, chunk=self.mlp_chunk,
                                                        slab_mask=slab_mask, shard_rows=True, tile_hint=tile_hint,
                                                        plain_slabs=plain)
        if plain:
            slab_mask = None
        virial = torch.empty((3, 3), dtype=torch.float64, device=dev) if stress else None
        pair_e, pair_g, pair_w = self._pair_terms(sp_l, x_l, cell, pbc_t, nbrs, lo, hi, stress)
        sae = self._sae64(dev) if self.energy_shifter._enabled else None
        e_atom = atomic_e if pair_e is None else atomic_e + pair_e
        fixed = self.deterministic_forces
        if fixed:
            rows = torch.zeros((nl, 3), dtype=torch.int64, device=dev)
            eng.backward(sp_l, nbrs, grad_aev, grad_coords=rows, shard_rows=True, virial=virial, slab_mask=slab_mask,
                         fixed_point=True)
            if pair_g is not None:
                rows += torch.round(pair_g.to(torch.float64) / FIXED_SCALE).to(torch.int64)
            energies = energy_reduce(sp_given, e_atom, sae, lo, hi)
            rows.neg_()
        else:
            rows = torch.zeros((nl, 3), dtype=torch.float32, device=dev)
            eng.backward(sp_l, nbrs, grad_aev, grad_coords=rows, shard_rows=True, virial=virial, slab_mask=slab_mask)
            if pair_g is not None:
                rows += pair_g
            energies = energy_forces_finish(sp_given, e_atom, sae, rows, lo, hi)   # (negates rows: forces)
        if stress and pair_w is not None:
            virial += pair_w
        tail = energies if not stress else torch.cat([energies, virial.reshape(-1)])
        from .parallel import FORCE_COLLECTIVES

        several = world > 1 or FORCE_COLLECTIVES
        if group is not None and several:
            if fixed:
                rows, tot = part.exchange(rows, torch.round(tail / FIXED_SCALE).to(torch.int64), group)
                tot = tot.to(torch.float64) * FIXED_SCALE
            else:
                rows, tot = part.exchange(rows, tail, group)
            energies = tot[:1].clone()
            if stress:
                virial = tot[1:].reshape(3, 3).clone()
            n_coll, nbytes = 1, part.last_bytes
        else:
            n_coll, nbytes = 0, 0
        f_l = fixed_to_float(rows) if fixed else rows
        if group is not None and several and reduce_forces:
            both = part.gather_owned(torch.cat([f_l, e_atom.view(-1, 1)], dim=1), group)   # ONE gather: forces + e_atom
            forces, ae = both[:, :3].contiguous(), both[:, 3].contiguous()
            n_coll, nbytes = n_coll + 1, nbytes + 16 * max(part.bounds[r + 1] - part.bounds[r] for r in range(world))
        elif group is None:   # shard=(rank, world) without a group: the rank's partial sums, halo pushes included
            forces, ae = part.scatter_local(f_l), part.scatter_owned(e_atom)
        else:
            forces, ae = part.scatter_owned(f_l), part.scatter_owned(e_atom)
        self.last_collective = {"collectives_per_step": n_coll, "world_size": world, "bytes": nbytes,
                                "op": "all_to_all(halo force rows -> slab neighbours, partial energy -> all)", "n_local": nl,
                                "peers": list(part.peers),
                                "n_owned": part.n_owned, "n_halo": part.n_left + part.n_right}
        if check_overflow:
            nbrs.raise_on_overflow()
        aevc._last_neighbors = nbrs
        return FusedEnergiesForces(energies, forces.view(1, n, 3), ae.view(1, n), virial)

    def _sae64(self, device) -> Tensor:
        """Self energies as float64 on ``device``, converted once per value of the buffer (a launch per step otherwise)."""
        src = self.energy_shifter.self_energies
        key = (src.data_ptr(), src._version, src.dtype, device)
        hit = self.__dict__.get("_sae64_cache")
        if hit is None or hit[0] != key:
            hit = (key, src.detach().to(device=device, dtype=torch.float64).clone())
            self.__dict__["_sae64_cache"] = hit
        return hit[1]

    def _pair_terms(self, species32: Tensor, c32: Tensor, cell, pbc, nbrs, lo: int, hi: int, stress: bool):
        """Pair-potential part of this rank's central atoms: (per-atom energies [N] or None, gradient [N, 3], virial)."""
        pots = [(k, p) for k, p in self.potentials.items() if k != "nnp" and p._enabled]
        if not pots:
            return None, None, None
        n = species32.numel()
        e = torch.zeros(n, dtype=torch.float32, device=c32.device)
        g = torch.zeros((n, 3), dtype=torch.float32, device=c32.device)
        w = torch.zeros((3, 3), dtype=torch.float64, device=c32.device) if stress else None
        for _, pot in pots:
            rows = nbrs
            if getattr(pot, "needs_all_rows", False):
                # (coordination numbers of every neighbor: rows of all atoms on every rank, outputs for lo .. hi only)
                rows = self._pair_rows(pot, species32, c32, cell, pbc, 0, None)
                pot.accumulate(species32, rows, e, g, w, lo=lo, hi=hi)
                continue
            if pot.cutoff > self.aev_computer.radial.cutoff + 1e-6:
                rows = self._pair_rows(pot, species32, c32, cell, pbc, lo, hi)
            pot.accumulate(species32, rows, e, g, w)
        return e, g, w

    def ase(self, overwrite: bool = False, stress_kind: str = "fdotr"):
        """ASE calculator for this model (arch.py ``ANI.ase``, torchani/ase.py:32-173)."""
        from .ase import Calculator

        return Calculator(self, overwrite=overwrite, stress_kind=stress_kind)

    # ---- external neighbor lists (arch.py:151-206,354-381) ------------------------------------------
    def compute_from_neighbors(self, elem_idxs: Tensor, coords: Tensor, neighbors, charge: int = 0,
                               atomic: bool = False, ensemble_values: bool = False) -> "EnergiesScalars":
        """Energies from element indices and an already screened half neighbor list (any
        (indices [2, P], distances [P], diff_vectors [P, 3]) tuple in the reference's convention).
        Differentiable with respect to ``coords`` like the reference's native cuAEV entry point."""
        assert charge == 0, "Model only supports neutral molecules"
        pair = [k for k, pot in self.potentials.items() if k != "nnp" and pot._enabled]
        if pair:
            # the reference loops over every enabled potential here (arch.py:353-381); the pair kernels of this package
            # run on the engine's own rows (D3 needs the rows of all atoms), not on an external half list
            raise NotImplementedError(f"compute_from_neighbors with enabled pair potentials {pair}: evaluate through "
                                      "forward / energies_and_forces, or set_enabled(name, False) first")
        energies = coords.new_zeros(elem_idxs.shape if atomic else elem_idxs.shape[:1])
        if ensemble_values:
            energies = energies.unsqueeze(0)
        if self.potentials["nnp"]._enabled:
            aevs = self.aev_computer.compute_from_neighbors(elem_idxs, coords, neighbors)
            energies = energies + self.neural_networks(elem_idxs, aevs, atomic, ensemble_values)
        if self.energy_shifter._enabled:
            energies = energies + self.energy_shifter(elem_idxs, atomic=atomic)
        return EnergiesScalars(energies)   # arch.py:353-381

    def compute_from_external_neighbors(self, species: Tensor, coords: Tensor, neighbor_idxs: Tensor,
                                        shifts: tp.Optional[Tensor], charge: int = 0, atomic: bool = False,
                                        ensemble_values: bool = False,
                                        _molecule_idxs: tp.Optional[Tensor] = None) -> "EnergiesScalars":
        """Entry point for a neighbor list owned by an MD engine: ``neighbor_idxs`` [2, P] and cartesian image
        ``shifts`` [P, 3] (or None); coords must be mapped to the central cell.  Pairs beyond the cutoff (Verlet
        skin) and pairs with padding atoms are dropped by the ingestion kernel (arch.py:171-206)."""
        elem_idxs = self._elem_idxs(species)
        if _molecule_idxs is not None:
            # experimental in the reference too (arch.py:194-203): pairs between different molecules of the one
            # conformation are dropped (discard_inter_molecule_pairs, neighbors.py:31-43)
            if coords.shape[0] != 1:
                raise ValueError("molecule_idxs expects only one conformation")
            if len(_molecule_idxs) != coords.shape[1]:
                raise ValueError("molecule_idxs must be the same length as num atoms, if passed")
            mol = _molecule_idxs.to(neighbor_idxs.device)
            keep = (mol[neighbor_idxs[0]] == mol[neighbor_idxs[1]]).nonzero().view(-1)
            neighbor_idxs = neighbor_idxs.index_select(1, keep)
            shifts = None if shifts is None else shifts.index_select(0, keep)
        flat = coords.detach().reshape(-1, 3)
        diff = flat.index_select(0, neighbor_idxs[0]) - flat.index_select(0, neighbor_idxs[1])
        if shifts is not None:
            diff = diff + shifts.to(diff.dtype)
        neighbors = (neighbor_idxs, diff.norm(dim=-1), diff)
        return self.compute_from_neighbors(elem_idxs, coords, neighbors, charge, atomic, ensemble_values)

    # ---- ensemble conveniences of the reference model (arch.py:133-135,245-264,385-585) ---------------
    def set_active_members(self, idxs: tp.Sequence[int]) -> None:
        self.neural_networks.set_active_members(list(idxs))

    def __len__(self) -> int:
        return self.neural_networks.get_active_members_num() if hasattr(self.neural_networks, "members") else 1

    def __getitem__(self, idx: int) -> "ANI":
        """Model with the idx-th ensemble member only (shares the parameters and the AEV computer)."""
        nets = self.neural_networks
        member = nets.members[idx] if hasattr(nets, "members") else nets
        m = ANI(self.symbols, self.aev_computer, member, self.energy_shifter.self_energies.tolist(),
                self.periodic_table_index)
        self._carry_over(m)
        return m.to(self.atomic_numbers.device)

    def _carry_over(self, m: "ANI") -> None:
        """What model[idx] keeps besides the member's networks: every other potential (shared modules) with its enabled
        flag, like the reference's deep copy of the whole model (arch.py:252-261)."""
        for name, pot in self.potentials.items():
            if name != "nnp":
                m.potentials[name] = pot
        m.potentials["nnp"]._enabled = self.potentials["nnp"]._enabled
        m.energy_shifter._enabled = self.energy_shifter._enabled
        for attr in ("mlp_chunk", "deterministic_forces", "auto_graph_atoms", "compact_species", "partition", "partition_skin", "partition_check", "locality_sort", "two_product_backward"):
            setattr(m, attr, getattr(self, attr))

    def atomic_energies(self, species_coordinates, cell=None, pbc=None, charge: int = 0,
                        ensemble_values: bool = False) -> SpeciesEnergies:
        """Per-atom energies [C, A] (or [M, C, A]), arch.py:385-400."""
        return self(species_coordinates, cell, pbc, charge, True, ensemble_values)

    def energies_qbcs(self, species_coordinates, cell=None, pbc=None, unbiased: bool = True,
                      charge: int = 0) -> SpeciesEnergiesQBC:
        """Ensemble-mean energies and query-by-committee factors std_m(E) / sqrt(n_atoms), arch.py:438-486."""
        elem_idxs, energies = self(species_coordinates, cell, pbc, charge, False, True)[:2]
        if energies.shape[0] == 1:
            qbc = torch.zeros_like(energies).squeeze(0)
        else:
            qbc = energies.std(0, unbiased=unbiased)
        num_atoms = (elem_idxs >= 0).sum(dim=1, dtype=energies.dtype)
        return SpeciesEnergiesQBC(elem_idxs, energies.mean(dim=0), qbc / num_atoms.sqrt())

    def atomic_stdev(self, species_coordinates, cell=None, pbc=None, charge: int = 0,
                     ensemble_values: bool = False, unbiased: bool = True) -> AtomicStdev:
        """Standard deviation of the atomic energies across the ensemble, arch.py:488-516."""
        elem_idxs, energies = self(species_coordinates, cell, pbc, charge, True, True)[:2]
        if energies.shape[0] == 1:
            stdev = torch.zeros_like(energies).squeeze(0)
        else:
            stdev = energies.std(0, unbiased=unbiased)
        if not ensemble_values:
            energies = energies.mean(0)
        return AtomicStdev(elem_idxs, energies, stdev)

    @torch.no_grad()
    def members_forces(self, species_coordinates, cell=None, pbc=None, charge: int = 0) -> SpeciesForces:
        """Energies [M, C] and forces [M, C, A, 3] of every active member (arch.py:403-436).  The engine
        differentiates the ensemble mean in one pass; per-member forces take one fused pass per member."""
        assert charge == 0, "Model only supports neutral molecules"
        species, coords = species_coordinates
        elem_idxs = self._elem_idxs(species)
        nets = self.neural_networks
        if not hasattr(nets, "members"):
            out = self.energies_and_forces(species, coords, cell, pbc)
            return SpeciesForces(elem_idxs, out.energies.unsqueeze(0), out.forces.unsqueeze(0))
        active = list(nets.active_members_idxs)
        es, fs = [], []
        try:
            for m in active:
                nets.set_active_members([m])
                out = self.energies_and_forces(species, coords, cell, pbc)
                es.append(out.energies.clone())
                fs.append(out.forces.clone())
        finally:
            nets.set_active_members(active)
        return SpeciesForces(elem_idxs, torch.stack(es), torch.stack(fs))

    def force_magnitudes(self, species_coordinates, cell=None, pbc=None,
                         ensemble_values: bool = False) -> ForceMagnitudes:
        """L2 norm of the members' atomic force vectors, averaged by default (arch.py:518-541)."""
        species, _, mf = self.members_forces(species_coordinates, cell, pbc)
        mags = mf.norm(dim=-1)
        return ForceMagnitudes(species, mags if ensemble_values else mags.mean(0))

    def force_qbc(self, species_coordinates, cell=None, pbc=None, ensemble_values: bool = False,
                  unbiased: bool = True) -> ForceStdev:
        """Mean force magnitudes with their relative std and range across the ensemble (arch.py:543-577)."""
        species, mags = self.force_magnitudes(species_coordinates, cell, pbc, True)
        eps = 1e-8
        mean_mags = mags.mean(0)
        if mags.shape[0] == 1:
            rel_std = torch.zeros_like(mags).squeeze(0)
            rel_range = torch.ones_like(mags).squeeze(0)
        else:
            rel_std = (mags.std(0, unbiased=unbiased) + eps) / (mean_mags + eps)
            rel_range = ((mags.max(dim=0).values - mags.min(dim=0).values) + eps) / (mean_mags + eps)
        return ForceStdev(species, mags if ensemble_values else mean_mags, rel_std, rel_range)

    def graphed(self, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None,
                pbc: tp.Optional[tp.Sequence[bool]] = None) -> "GraphedEnergiesForces":
        """Capture energies_and_forces for this (species, shapes, pbc) into a HIP graph and return a callable
        ``f(coords, cell=None) -> FusedEnergiesForces`` that replays it: one graph launch instead of a dozen kernel
        launches, for launch-bound sizes (batches of small molecules, MD of small systems)."""
        return GraphedEnergiesForces(self, species, coords, cell, pbc)

    def load_reference_state_dict(self, state: tp.Mapping[str, tp.Any], strict: bool = False):
        """Load a (reference or seeded) state dict given as tensors or numpy arrays."""
        conv = {k: (torch.from_numpy(np.asarray(v)) if not isinstance(v, Tensor) else v) for k, v in state.items()}
        return self.load_state_dict(conv, strict=strict)


class ANIq(ANI):
    """ANI-style model that also predicts atomic charges from separate charge networks on the same AEVs (arch.py:579-692
    ANIq with SeparateChargesNNPotential, potentials/nnp.py:75-102).  ``forward`` returns SpeciesEnergiesAtomicCharges;
    the charges are the charge networks' outputs passed through the normalizer."""

    def __init__(self, symbols, aev_computer, neural_networks, self_energies, periodic_table_index: bool = True,
                 charge_networks: tp.Optional[torch.nn.Module] = None,
                 charge_normalizer: tp.Optional[torch.nn.Module] = None) -> None:
        super().__init__(symbols, aev_computer, neural_networks, self_energies, periodic_table_index)
        if charge_networks is None:
            raise ValueError("ANIq needs charge_networks (merged charge / energy networks are not implemented)")
        self.potentials["nnp"].charge_networks = charge_networks
        self.potentials["nnp"].charge_normalizer = (ChargeNormalizer(symbols) if charge_normalizer is None
                                                    else charge_normalizer)

    def __getitem__(self, idx: int) -> "ANIq":
        nets = self.neural_networks
        member = nets.members[idx] if hasattr(nets, "members") else nets
        nnp = self.potentials["nnp"]
        m = ANIq(self.symbols, self.aev_computer, member, self.energy_shifter.self_energies.tolist(),
                 self.periodic_table_index, nnp.charge_networks, nnp.charge_normalizer)
        self._carry_over(m)
        return m.to(self.atomic_numbers.device)

    def _scalars_from_aevs(self, elem_idxs: Tensor, aevs: Tensor, charge: int) -> Tensor:
        # potentials/nnp.py:99-102.  The GELU networks run the inference kernels only: the charges carry no gradient
        nnp = self.potentials["nnp"]
        with torch.no_grad():
            qs = nnp.charge_networks(elem_idxs, aevs.detach(), atomic=True)
            return nnp.charge_normalizer(elem_idxs, qs, charge)

    def _output(self, elem_idxs: Tensor, energies: Tensor, extra: tp.Optional[Tensor]):
        from .tuples import SpeciesEnergiesAtomicCharges

        qs = energies.new_zeros(elem_idxs.shape) if extra is None else extra.to(energies.dtype)
        return SpeciesEnergiesAtomicCharges(elem_idxs, energies, qs)

    def atomic_charges(self, species_coordinates, cell: tp.Optional[Tensor] = None,
                       pbc: tp.Optional[Tensor] = None, charge: int = 0) -> Tensor:
        """Normalized atomic charges [C, A] (padding atoms: 0)."""
        with torch.no_grad():
            return self(species_coordinates, cell, pbc, charge).atomic_charges


class GraphedEnergiesForces:
    """HIP-graph replay of ANI.energies_and_forces for fixed species / shapes (single process, no sharding).

    The kernels are launched through the C ABI on torch's current stream, so ``torch.cuda.graph`` records
    them like any other stream work; all buffers come from the graph's private memory pool.  Outputs are static
    tensors overwritten by every call (clone them to keep a result).  The graph OWNS what it points at: it keeps
    the packed weight planes it was captured with alive and pins their workspace (PackedNetworks.pinned), and it
    re-captures itself when the model's parameters (or active members) have changed since.  Neighbor-row overflow
    cannot raise inside a graph: call ``check()`` when convenient."""

    def __init__(self, model: ANI, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None,
                 pbc: tp.Optional[tp.Sequence[bool]] = None, warmup: int = 3) -> None:
        if not coords.is_cuda:
            raise ValueError("torchani_amd's engine needs tensors on a ROCm device (no CPU fallback)")
        self.model = model
        self.species32 = model._elem_idxs(species).to(torch.int32).contiguous()   # (validity check syncs once)
        self.coords = coords.detach().to(torch.float32).contiguous().clone()
        self.cell = None if cell is None else cell.detach().clone()
        self.pbc = pbc
        self.warmup = warmup
        self.tile_hint = model._tile_hint(self.species32, self.species32, self.species32.numel())
        # the kernels' species numbering (ANI.compact_species): worked out here, outside the capture, and OWNED by this
        # object like every other tensor the graph reads
        sp_e, self.species_order = model._engine_species(self.species32)
        self.engine_species32 = sp_e if self.species_order is None else sp_e.clone()
        self.n_captures = 0
        self._packed = None
        self._capture()

    def _capture(self) -> None:
        self._release()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):   # packs the weights, sizes the workspaces
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        self._packed = self.model.neural_networks._pack(self.coords.device, self.species_order)   # strong reference: planes + workspace
        self._packed.pinned += 1
        self._sae = self._current_sae()   # (the float64 copy of the self energies the graph reads)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = self._run()
        self.n_captures += 1

    def _current_sae(self) -> tp.Optional[Tensor]:
        m = self.model
        return m._sae64(self.coords.device) if m.energy_shifter._enabled else None

    def _release(self) -> None:
        if self._packed is not None:
            self._packed.pinned -= 1
            self._packed = None

    def __del__(self):
        self._release()

    def _run(self) -> FusedEnergiesForces:
        return self.model._energies_and_forces_core(self.species32, self.coords, self.cell, self.pbc, None, True,
                                                    False, None, tile_hint=self.tile_hint,
                                                    engine_species=(self.engine_species32, self.species_order))

    def __call__(self, coords: Tensor, cell: tp.Optional[Tensor] = None) -> FusedEnergiesForces:
        if (self.model.neural_networks._pack(self.coords.device, self.species_order) is not self._packed
                or self._current_sae() is not self._sae):
            self._capture()   # parameters were updated in place / other members: the old planes are stale
        self.coords.copy_(coords)
        if cell is not None:
            assert self.cell is not None, "the graph was captured without a cell"
            self.cell.copy_(cell)
        self.graph.replay()
        return self.out

    def check(self) -> None:
        """Raise if a neighbor row overflowed in the last replay (host sync)."""
        self.model.aev_computer.last_neighbors().raise_on_overflow()
        self.model._raise_on_pair_overflow()


def _assemble(kind: str, n_members: int, neighborlist: str, row_capacity: int,
              periodic_table_index: bool, cutoff_fn: str = "cosine") -> ANI:
    symbols, consts, hidden = arch_spec(kind)
    aevc = AEVComputer(consts, neighborlist=neighborlist, row_capacity=row_capacity, cutoff_fn=cutoff_fn)
    act, bias = arch_networks(kind)
    members = [ANINetworks.build(symbols, consts.out_dim, hidden, act, bias) for _ in range(n_members)]
    nets: torch.nn.Module = Ensemble(members) if n_members > 1 else members[0]
    sae = [arch_gsaes(kind)[s] for s in symbols]
    model = ANI(symbols, aevc, nets, sae, periodic_table_index)
    if kind in ("ani2xr", "ani2dr") or kind.startswith("anir2s"):
        # arch.py:1055-1066: repulsion up to the radial cutoff (ANI-r2s: without a cutoff), dispersion up to 8 A
        from .potentials import RepulsionXTB, TwoBodyDispersionD3

        rep_cut = math.inf if kind.startswith("anir2s") else consts.Rcr
        model.add_pair_potential("repulsion_xtb", RepulsionXTB(symbols, cutoff=rep_cut, cutoff_fn="smooth"))
        if kind == "ani2dr":
            model.add_pair_potential("dispersion_d3", TwoBodyDispersionD3.from_functional(
                symbols, "b973c", cutoff=8.0, cutoff_fn="smooth"))
    return model


def _builtin(kind: str, state_dict, seed, n_members, device, neighborlist, row_capacity,
             periodic_table_index, cutoff_fn: str = "cosine", model_index: tp.Optional[int] = None,
             strategy: str = "hip", dtype=None) -> ANI:
    if strategy not in ("hip", "auto", "pyaev", "cuaev", "cuaev-fused", "cuaev-interface"):
        raise ValueError(f"Unsupported strategy {strategy!r}")   # (every reference strategy maps to the HIP engine)
    if dtype not in (None, torch.float32):
        raise ValueError("the HIP engine computes in float32 (energies are reduced in float64)")
    model = _assemble(kind, n_members, neighborlist, row_capacity, periodic_table_index, cutoff_fn)
    if state_dict is None:
        # the published parameters are a download in the reference (arch.py:1185-1220) and are not shipped here: an
        # explicit seed gives the same architecture with seeded random parameters (tests, benchmarks)
        if seed is None:
            warnings.warn(f"torchani_amd.{kind.upper().replace('ANI', 'ANI')}: neither state_dict nor seed given -- the "
                          "model gets RANDOM parameters (seed 0), its energies and forces are meaningless. Pass "
                          "state_dict=<the trained reference parameters> (torchani.models.ANI2x().state_dict()).",
                          UserWarning, stacklevel=3)
        state_dict = random_state_dict(kind, n_members, 0 if seed is None else seed)
    if n_members == 1:
        # a single network is not wrapped in an Ensemble (no "members.0." level in its keys): accept the one-member
        # form of an ensemble state dict too
        pre = "potentials.nnp.neural_networks."
        state_dict = {(pre + k[len(pre) + len("members.0."):] if k.startswith(pre + "members.0.") else k): v
                      for k, v in state_dict.items()}
    res = model.load_reference_state_dict(state_dict, strict=False)
    lost = [k for k in res.missing_keys if k.startswith(("potentials.nnp.neural_networks", "energy_shifter"))]
    if lost:
        raise RuntimeError(f"state_dict does not provide {len(lost)} network / self-energy tensors (first: {lost[0]}): "
                           "it is not a state dict of this architecture (a Bmm / infer-converted reference model "
                           "must be saved before conversion); refusing to run on random weights")
    model.requires_grad_(False)
    if device is not None:
        model = model.to(device)
    # models.py:195: a single member of the ensemble on request
    return model if model_index is None else model[model_index]


def ANI2x(model_index: tp.Optional[int] = None, neighborlist: str = "auto", strategy: str = "hip",
          periodic_table_index: bool = True, device=None, dtype=None, state_dict=None,
          seed: tp.Optional[int] = None, n_members: int = 8, row_capacity: int = 128,
          cutoff_fn: str = "cosine") -> ANI:
    """ANI-2x architecture: H C N O S F Cl, 1008-dim AEV, 8-member ensemble (models.py:185-196).
    cutoff_fn="smooth" gives the envelope of the reference's newer models (arch.py:1006, CutoffSmooth)."""
    return _builtin("ani2x", state_dict, seed, n_members, device, neighborlist, row_capacity,
                    periodic_table_index, cutoff_fn, model_index, strategy, dtype)


def ANI2xr(model_index: tp.Optional[int] = None, neighborlist: str = "auto", strategy: str = "hip",
           periodic_table_index: bool = True, device=None, dtype=None, state_dict=None,
           seed: tp.Optional[int] = None, n_members: int = 8, row_capacity: int = 128) -> ANI:
    """ANI-2xr architecture (models.py:252-287): H C N O F S Cl, the AEV of simple_ani (smooth envelope, radial cutoff
    5.2 A), GELU networks without biases, xTB repulsion."""
    return _builtin("ani2xr", state_dict, seed, n_members, device, neighborlist, row_capacity,
                    periodic_table_index, "smooth", model_index, strategy, dtype)


def ANI2dr(model_index: tp.Optional[int] = None, neighborlist: str = "auto", strategy: str = "hip",
           periodic_table_index: bool = True, device=None, dtype=None, state_dict=None,
           seed: tp.Optional[int] = None, n_members: int = 8, row_capacity: int = 128) -> ANI:
    """ANI-2dr architecture (models.py:290-325): ANI-2xr + DFT-D3(BJ) dispersion (B97-3c constants, 8 A), self energies
    of B97-3c / def2-mTZVP."""
    return _builtin("ani2dr", state_dict, seed, n_members, device, neighborlist, row_capacity,
                    periodic_table_index, "smooth", model_index, strategy, dtype)


def ANIr2s(model_index: tp.Optional[int] = None, neighborlist: str = "auto", strategy: str = "hip",
           periodic_table_index: bool = True, device=None, dtype=None, solvent: tp.Optional[str] = None,
           state_dict=None, seed: tp.Optional[int] = None, n_members: int = 8, row_capacity: int = 128) -> ANI:
    """ANI-r2s architecture (models.py:325-368; solvent = None, "water", "chcl3" or "ch3cn" picks the self energies):
    the ANI-2x AEV with the smooth envelope, GELU networks without biases, xTB repulsion WITHOUT a cutoff (molecules
    only: the repulsion rows hold every pair of a molecule, at most 256 neighbors per atom)."""
    if solvent not in (None, "water", "chcl3", "ch3cn"):
        raise ValueError(f"unknown solvent {solvent!r}")
    return _builtin("anir2s" + ("" if solvent is None else "_" + solvent), state_dict, seed, n_members, device,
                    neighborlist, row_capacity, periodic_table_index, "smooth", model_index, strategy, dtype)


def ANIr2s_water(**kw) -> ANI:
    return ANIr2s(solvent="water", **kw)


def ANIr2s_chcl3(**kw) -> ANI:
    return ANIr2s(solvent="chcl3", **kw)


def ANIr2s_ch3cn(**kw) -> ANI:
    return ANIr2s(solvent="ch3cn", **kw)


def ANImbis(model_index: tp.Optional[int] = None, neighborlist: str = "auto", strategy: str = "hip",
            periodic_table_index: bool = True, device=None, dtype=None, state_dict=None,
            seed: tp.Optional[int] = None, n_members: int = 8, row_capacity: int = 128) -> ANIq:
    """ANI-mbis architecture (models.py:201-252): ANI-2x energies + MBIS atomic charges from one set of GELU / bias-free
    charge networks with two outputs (the second is the charge), normalized with electronegativity / hardness weights
    scaled by the squared raw charges."""
    from .nn import ANINetworksDiscardFirstScalar
    from .weights import random_charge_state_dict

    base = _builtin("ani2x", state_dict, seed, n_members, None, neighborlist, row_capacity, periodic_table_index,
                    "cosine", None, strategy, dtype)
    symbols, consts, hidden = arch_spec("ani2x")
    qnets = ANINetworksDiscardFirstScalar.build(symbols, consts.out_dim, hidden, "gelu", False, out_dim=2)
    model = ANIq(symbols, base.aev_computer, base.neural_networks,
                 [float(v) for v in base.energy_shifter.self_energies], periodic_table_index, qnets,
                 ChargeNormalizer.from_electronegativity_and_hardness(symbols, scale_weights_by_charges_squared=True))
    sd = state_dict if state_dict is not None else {}
    pre = "potentials.nnp.charge_networks."
    qsd = {k[len(pre):]: (torch.from_numpy(np.asarray(v)) if not isinstance(v, Tensor) else v)
           for k, v in sd.items() if k.startswith(pre)}
    if not qsd:   # (seeded stand-in for the reference's charge_nn_state_dict.pt download)
        qsd = {k: torch.from_numpy(v) for k, v in random_charge_state_dict(0 if seed is None else seed).items()}
    qnets.load_state_dict(qsd, strict=True)
    model.requires_grad_(False)
    if device is not None:
        model = model.to(device)
    return model if model_index is None else model[model_index]


def simple_ani(symbols: tp.Sequence[str], lot: str, ensemble_size: int = 1, radial_start: float = 0.9,
               angular_start: float = 0.9, radial_cutoff: float = 5.2, angular_cutoff: float = 3.5,
               radial_shifts: int = 16, angular_shifts: int = 8, sections: int = 4, radial_precision: float = 19.7,
               angular_precision: float = 12.5, angular_zeta: float = 14.1, cutoff_fn: str = "smooth",
               dispersion: bool = False, repulsion: bool = True, container_ctor: str = "default",
               container: str = "ANINetworks", activation: str = "gelu", bias: bool = False, strategy: str = "auto",
               periodic_table_index: bool = True, neighborlist: str = "auto", repulsion_cutoff: bool = True,
               seed: tp.Optional[int] = None, state_dict=None, device=None, row_capacity: int = 128) -> ANI:
    """The reference's flexible builder (arch.py:992-1066) on the HIP engine: symmetry functions that cover the radial /
    angular range linearly (ANIRadial / ANIAngular.cover_linearly, aev/_terms.py:189-207,346-366), ANINetworks of the
    ANI-2x ("default", "like_2x") or ANI-1x ("like_1x") widths, self energies of the level of theory ``lot``
    (constants.GSAES), optionally the xTB repulsion and the D3 dispersion of that functional.  Like the reference's, the
    networks start from random parameters (``seed`` makes them reproducible; ``state_dict`` loads trained ones).  CELU
    networks with biases come back trainable (the training passes cover those); GELU / bias-free ones are frozen.

    What the kernels cover (a ValueError names anything else): 16 radial shifts, an 8 x 4 or 4 x 8 angular grid (shifts x
    sections), at most 7 elements, three hidden layers of at most 256 units, CELU (with biases: trainable here) or GELU."""
    from .constants import ATOMIC_NUMBER, GSAES, HIDDEN_DIMS_1X, HIDDEN_DIMS_2X, AEVConstants, linspace
    from .weights import random_network_state_dict

    symbols = tuple(symbols)
    if not 1 <= len(symbols) <= 7 or any(s not in ATOMIC_NUMBER for s in symbols) or len(set(symbols)) != len(symbols):
        raise ValueError(f"simple_ani: 1 to 7 distinct elements out of {sorted(ATOMIC_NUMBER)}, got {symbols}")
    if radial_shifts != 16 or (angular_shifts, sections) not in ((8, 4), (4, 8)):
        raise ValueError("the AEV kernels cover radial_shifts=16 with angular_shifts x sections = 8 x 4 or 4 x 8 "
                         f"(got {radial_shifts}, {angular_shifts} x {sections})")
    if container != "ANINetworks" or container_ctor not in ("default", "like_2x", "like_1x"):
        raise ValueError("the network kernels cover container='ANINetworks' with container_ctor 'default' / 'like_2x' / "
                         f"'like_1x' (got {container!r}, {container_ctor!r}): SingleNN and shared-layer containers are not "
                         "implemented")
    if activation not in ("celu", "gelu"):
        raise ValueError(f"activation 'celu' or 'gelu', got {activation!r}")
    from .constants import cutoff_kernel_name as kernel_name

    cutoff_fn = kernel_name(cutoff_fn)
    if cutoff_fn not in ("cosine", "smooth"):
        raise ValueError(f"cutoff_fn 'cosine' or 'smooth', got {cutoff_fn!r}")
    if strategy not in ("hip", "auto", "pyaev", "cuaev", "cuaev-fused", "cuaev-interface"):
        raise ValueError(f"Unsupported strategy {strategy!r}")
    if lot.lower() not in GSAES:
        raise KeyError(f"no ground-state atomic energies for the level of theory {lot!r}: {sorted(GSAES)}")
    gsaes = GSAES[lot.lower()]
    angle_start = math.pi / sections / 2
    consts = AEVConstants(len(symbols), float(radial_cutoff), float(angular_cutoff), float(radial_precision),
                          linspace(radial_start, radial_cutoff, radial_shifts), float(angular_precision),
                          float(angular_zeta), linspace(angular_start, angular_cutoff, angular_shifts),
                          linspace(angle_start, math.pi + angle_start, sections), cutoff_fn)
    # nn/_containers.py:479-544: per-element widths, (160, 128, 96) / (128, 112, 96) for elements without an entry
    table, other = (HIDDEN_DIMS_1X, (128, 112, 96)) if container_ctor == "like_1x" else (HIDDEN_DIMS_2X, (160, 128, 96))
    hidden = {s: table.get(s, other) for s in symbols}
    aevc = AEVComputer(consts, neighborlist=neighborlist, row_capacity=row_capacity, cutoff_fn=cutoff_fn)
    members = [ANINetworks.build(symbols, consts.out_dim, hidden, activation, bias) for _ in range(ensemble_size)]
    nets: torch.nn.Module = Ensemble(members) if ensemble_size > 1 else members[0]
    model = ANI(symbols, aevc, nets, [gsaes[s] for s in symbols], periodic_table_index)
    if repulsion:
        from .potentials import RepulsionXTB

        model.add_pair_potential("repulsion_xtb", RepulsionXTB(
            symbols, cutoff=float(radial_cutoff) if repulsion_cutoff else math.inf, cutoff_fn="smooth"))
    if dispersion:
        from .potentials import TwoBodyDispersionD3

        model.add_pair_potential("dispersion_d3", TwoBodyDispersionD3.from_functional(
            symbols, lot.split("-")[0], cutoff=8.0, cutoff_fn="smooth"))
    if state_dict is not None:
        res = model.load_reference_state_dict(state_dict, strict=False)
        lost = [k for k in res.missing_keys if k.startswith("potentials.nnp.neural_networks")]
        if lost:
            raise RuntimeError(f"state_dict does not provide {len(lost)} network tensors (first: {lost[0]})")
    elif seed is not None:
        model.load_reference_state_dict(random_network_state_dict(symbols, consts.out_dim, hidden, ensemble_size, seed,
                                                                  bias), strict=False)
    if activation != "celu" or not bias:
        model.requires_grad_(False)   # (the training passes cover CELU networks with biases: the others are inference models)
    if device is not None:
        model = model.to(device)
    return model


def simple_aniq(symbols: tp.Sequence[str], lot: str, ensemble_size: int = 1, radial_start: float = 0.9,
                angular_start: float = 0.9, radial_cutoff: float = 5.2, angular_cutoff: float = 3.5,
                radial_shifts: int = 16, angular_shifts: int = 8, sections: int = 4, radial_precision: float = 19.7,
                angular_precision: float = 12.5, angular_zeta: float = 14.1, cutoff_fn: str = "smooth",
                dispersion: bool = False, repulsion: bool = True, container_ctor: str = "default",
                charge_container_ctor: str = "default", container: str = "ANINetworks",
                charge_container: str = "ANINetworks", activation: str = "gelu", bias: bool = False,
                strategy: str = "auto", merge_charge_networks: bool = False,
                scale_charge_normalizer_weights: bool = True, dummy_energies: bool = False, use_cuda_ops: bool = False,
                periodic_table_index: bool = True, neighborlist: str = "auto", normalize: bool = True,
                seed: tp.Optional[int] = None, state_dict=None, device=None, row_capacity: int = 128) -> ANIq:
    """The reference's flexible builder for models that also output atomic charges (arch.py:1069-1185), with SEPARATE charge
    networks (one more set of ANINetworks on the same AEVs) and the electronegativity / hardness normalizer
    (``normalize=False``: raw charges).  ``merge_charge_networks`` and ``dummy_energies`` are not implemented."""
    from .constants import HIDDEN_DIMS_1X, HIDDEN_DIMS_2X
    from .weights import NN_PREFIX, random_network_state_dict

    if merge_charge_networks or dummy_energies:
        raise ValueError("simple_aniq: merge_charge_networks / dummy_energies are not implemented (separate charge networks "
                         "next to real energy networks only)")
    if charge_container != "ANINetworks" or charge_container_ctor not in ("default", "like_2x", "like_1x"):
        raise ValueError("charge_container='ANINetworks' with charge_container_ctor 'default' / 'like_2x' / 'like_1x' "
                         f"(got {charge_container!r}, {charge_container_ctor!r})")
    base = simple_ani(symbols, lot, ensemble_size, radial_start, angular_start, radial_cutoff, angular_cutoff, radial_shifts,
                      angular_shifts, sections, radial_precision, angular_precision, angular_zeta, cutoff_fn, dispersion,
                      repulsion, container_ctor, container, activation, bias, strategy, periodic_table_index, neighborlist,
                      True, seed, None, None, row_capacity)
    symbols = tuple(symbols)
    in_dim = base.aev_computer.constants().out_dim
    table, other = ((HIDDEN_DIMS_1X, (128, 112, 96)) if charge_container_ctor == "like_1x"
                    else (HIDDEN_DIMS_2X, (160, 128, 96)))
    hidden = {s: table.get(s, other) for s in symbols}
    qnets = ANINetworks.build(symbols, in_dim, hidden, activation, bias)
    normalizer = (ChargeNormalizer.from_electronegativity_and_hardness(
        symbols, scale_weights_by_charges_squared=scale_charge_normalizer_weights) if normalize else BaseChargeNormalizer())
    model = ANIq(symbols, base.aev_computer, base.neural_networks, [float(v) for v in base.energy_shifter.self_energies],
                 periodic_table_index, qnets, normalizer)
    for name, pot in base.potentials.items():
        if name != "nnp":
            model.add_pair_potential(name, pot)
    qpre = "potentials.nnp.charge_networks."
    if state_dict is not None:
        res = model.load_reference_state_dict(state_dict, strict=False)
        lost = [k for k in res.missing_keys if k.startswith(("potentials.nnp.neural_networks", qpre))]
        if lost:
            raise RuntimeError(f"state_dict does not provide {len(lost)} network tensors (first: {lost[0]})")
    elif seed is not None:
        q = random_network_state_dict(symbols, in_dim, hidden, 1, 1000 + seed, bias)
        model.load_reference_state_dict({qpre + k[len(NN_PREFIX):]: v for k, v in q.items()}, strict=False)
    if activation != "celu" or not bias:
        model.requires_grad_(False)
    else:
        qnets.requires_grad_(False)   # (charges come from the inference kernels, whatever the networks)
    if device is not None:
        model = model.to(device)
    return model


def ANI1ccx(model_index: tp.Optional[int] = None, neighborlist: str = "auto", strategy: str = "hip",
            periodic_table_index: bool = True, device=None, dtype=None, state_dict=None,
            seed: tp.Optional[int] = None, n_members: int = 8, row_capacity: int = 128,
            cutoff_fn: str = "cosine") -> ANI:
    """ANI-1ccx architecture (models.py:128-162): ANI-1x networks and AEV, CCSD(T)*/CBS self energies."""
    return _builtin("ani1ccx", state_dict, seed, n_members, device, neighborlist, row_capacity,
                    periodic_table_index, cutoff_fn, model_index, strategy, dtype)


def ANI1x(model_index: tp.Optional[int] = None, neighborlist: str = "auto", strategy: str = "hip",
          periodic_table_index: bool = True, device=None, dtype=None, state_dict=None,
          seed: tp.Optional[int] = None, n_members: int = 8, row_capacity: int = 128,
          cutoff_fn: str = "cosine") -> ANI:
    """ANI-1x architecture: H C N O, 384-dim AEV, 8-member ensemble (models.py:112-119)."""
    return _builtin("ani1x", state_dict, seed, n_members, device, neighborlist, row_capacity,
                    periodic_table_index, cutoff_fn, model_index, strategy, dtype)
