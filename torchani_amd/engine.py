"""Tensor-level front-end of the C ABI: torch tensors in, torch tensors out, current HIP stream.

PyTorch is used here only as plumbing (device memory, streams); all arithmetic of the hot path happens
inside libanihip.so.  Every function requires CUDA(ROCm) tensors and raises otherwise -- there is no
eager/CPU fallback on purpose.
"""
from __future__ import annotations

import ctypes as C
import typing as tp

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .constants import AEVConstants


def _ptr(t: tp.Optional[Tensor]) -> tp.Optional[int]:
    return None if t is None else t.data_ptr()


def _row_ptr(t: Tensor, row_offset: int, row_len: int) -> int:
    """Base pointer of a [*, row_len] fp32 buffer whose first row is atom ``row_offset``: the C ABI indexes
    rows by absolute atom index but only touches the central range, so a rank can keep buffers of its shard
    alone (the shifted base is never dereferenced outside the shard's rows)."""
    return t.data_ptr() - 4 * row_offset * row_len


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(*ts: tp.Optional[Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            # same error class as the reference for non-CUDA input to the native path
            # (aev/_computer.py:444-447)
            raise ValueError("torchani_amd's HIP engine needs tensors on a ROCm device (no CPU fallback)")


class NeighborRows(tp.NamedTuple):
    """Fixed-capacity, species-sorted full neighbor rows (format: include/anihip.h)."""

    meta: Tensor      # [N, 6] int32 (uint32 words)
    ent: Tensor       # [(hi-lo) * row_cap, 4] float32
    status: Tensor    # [8] int32
    row_cap: int
    lo: int
    hi: int
    symmetric: bool = True   # j in row i <=> i in row j (false for LAMMPS full lists with ghost atoms)

    def overflowed(self) -> bool:
        """Synchronising read of the device-side status word: did any row exceed its capacity (and was zeroed)?"""
        return bool(int(self.status[0].item()) & (_lib.ST_ROW_OVERFLOW | _lib.ST_ENTRY_OVERFLOW))

    def raise_on_overflow(self) -> None:
        """Synchronising check of the device-side status words."""
        st = int(self.status[0].item())
        if st & (_lib.ST_ROW_OVERFLOW | _lib.ST_ENTRY_OVERFLOW):
            raise RuntimeError(
                f"neighbor row overflow (status={st}): an atom has more than row_capacity={self.row_cap} "
                f"neighbors within the radial cutoff, > {_lib.MAX_ANG} within the angular cutoff or > 255 of "
                "one species; raise row_capacity (max 256)")


def rows_to_half(nbrs: NeighborRows, n_atoms: int) -> tp.Tuple[Tensor, Tensor, Tensor]:
    """Neighbor rows -> the reference's half list (indices [2, P] int64, distances [P], diff_vectors [P, 3] = r_i - r_j
    (+ shift)): what ``torch.ops.cell_list.cell_list`` / ``FastCellList`` return (csrc/cell_list.cpp:342-354,
    neighbors.py:285-294).  anihip_nbr_rows_to_half, two calls: count, then write (one host sync for P)."""
    L = _lib.lib()
    dev = nbrs.meta.device
    ws = torch.empty(L.anihip_nbr_rows_to_half_workspace_bytes(nbrs.hi - nbrs.lo), dtype=torch.uint8, device=dev)
    npairs = torch.zeros(1, dtype=torch.int64, device=dev)
    args = (_stream(), n_atoms, nbrs.lo, nbrs.hi, _ptr(nbrs.meta), _ptr(nbrs.ent), _ptr(ws), ws.numel())
    _lib.check(L.anihip_nbr_rows_to_half(*args, 0, None, None, None, _ptr(npairs)))
    P = int(npairs.item())
    idx = torch.empty((2, P), dtype=torch.int64, device=dev)
    dist = torch.empty(P, dtype=torch.float32, device=dev)
    diff = torch.empty((P, 3), dtype=torch.float32, device=dev)
    if P:
        _lib.check(L.anihip_nbr_rows_to_half(*args, P, _ptr(idx), _ptr(dist), _ptr(diff), _ptr(npairs)))
    return idx, dist, diff


class AevEngine:
    """Neighbor rows + AEV forward/backward for one set of AEV constants."""

    def __init__(self, consts: AEVConstants) -> None:
        self.consts = consts
        p = _lib.AevParams()
        p.num_species = consts.num_species
        p.n_shf_r, p.n_shf_a, p.n_shf_z = len(consts.ShfR), len(consts.ShfA), len(consts.ShfZ)
        p.Rcr, p.Rca = consts.Rcr, consts.Rca
        p.EtaR, p.EtaA, p.Zeta = consts.EtaR, consts.EtaA, consts.Zeta
        if consts.cutoff_fn not in _lib.CUTOFF_KINDS:
            raise ValueError(f"Unsupported cutoff function {consts.cutoff_fn!r}: the HIP kernels have "
                             f"{sorted(_lib.CUTOFF_KINDS)}")
        p.cutoff_kind = _lib.CUTOFF_KINDS[consts.cutoff_fn]
        if not (1 <= p.n_shf_r <= 32 and 1 <= p.n_shf_a <= 16 and 1 <= p.n_shf_z <= 16 and 1 <= p.num_species <= 7):
            raise ValueError(
                "HIP AEV kernels support up to 32 radial shifts, angular grids up to 16 x 16 and up to 7 species; "
                f"got nR={p.n_shf_r}, nA x nZ={p.n_shf_a}x{p.n_shf_z}, S={p.num_species}")
        # the published grids (16 radial shifts, 8x4 = ANI-2x or 4x8 = ANI-1x angular terms) run through the tuned kernels
        # with slab masks and the forward-mode derivative; any other grid through the general kernels (csrc/aev_generic.hip)
        self.tuned = p.n_shf_r == 16 and (p.n_shf_a, p.n_shf_z) in ((8, 4), (4, 8))
        self.params = p
        self.L = consts.out_dim
        self._host_table: tp.Optional[np.ndarray] = None
        self._tables: tp.Dict[torch.device, Tensor] = {}

    def host_table(self) -> np.ndarray:
        if self._host_table is None:
            c = self.consts
            shfr = np.asarray(c.ShfR, dtype=np.float32)
            shfa = np.asarray(c.ShfA, dtype=np.float32)
            shfz = np.asarray(c.ShfZ, dtype=np.float32)
            out = np.zeros(_lib.TABLE_FLOATS, dtype=np.float32)
            _lib.check(_lib.lib().anihip_aev_table_pack(
                C.byref(self.params), shfr.ctypes.data, shfa.ctypes.data, shfz.ctypes.data, out.ctypes.data))
            self._host_table = out
        return self._host_table

    def table(self, device: torch.device) -> Tensor:
        t = self._tables.get(device)
        if t is None:
            t = torch.from_numpy(self.host_table()).to(device)
            self._tables[device] = t
        return t

    # ---- neighbor rows ----------------------------------------------------------------------------
    def neighbors(self, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor] = None,
                  pbc: tp.Optional[tp.Sequence[bool]] = None, lo: int = 0, hi: tp.Optional[int] = None,
                  mode: str = "auto", row_cap: int = 128, max_cells: tp.Optional[int] = None) -> NeighborRows:
        """species [C,A] int32, coords [C,A,3] float32 (contiguous).  mode: batch | cell | auto."""
        _require_cuda(species, coords, cell)
        assert species.dtype == torch.int32 and coords.dtype == torch.float32
        assert species.is_contiguous() and coords.is_contiguous()
        Cn, A = species.shape
        n = Cn * A
        hi = n if hi is None else hi
        dev = coords.device
        pbc_mask = 0
        cell_t = None
        if cell is not None and pbc is not None and any(bool(b) for b in pbc):
            cell_t = cell.detach().to(device=dev, dtype=torch.float32).contiguous()
            pbc_mask = sum((1 << k) for k in range(3) if bool(pbc[k]))
        if mode == "auto":
            # the reference switches from all-pairs to the cell list at 190 (pbc) / 1770 atoms
            # (neighbors.py:324); a wave sweeps 64 candidates at a time so the crossover is later here
            mode = "cell" if (Cn == 1 and A > 512) else "batch"
        if mode == "cell" and Cn != 1:
            raise ValueError("the cell-list builder handles one system at a time (neighbors.py:373-381)")
        row_cap = int(min(max(row_cap, 1), _lib.MAX_RAD))
        n_central = max(hi - lo, 0)
        meta = torch.empty((n, _lib.META_WORDS), dtype=torch.int32, device=dev)
        ent = torch.empty((max(n_central, 1) * row_cap, 4), dtype=torch.float32, device=dev)
        # (the two builders reset the status words in their first kernel, include/anihip.h; an empty range returns early)
        status = (torch.empty if n_central > 0 else torch.zeros)(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        L = _lib.lib()
        if mode == "batch":
            ws = torch.empty(L.anihip_nbr_workspace_bytes(n, 1), dtype=torch.uint8, device=dev)
            _lib.check(L.anihip_nbr_build_batch(
                _stream(), C.byref(self.params), Cn, A, _ptr(species), _ptr(coords), _ptr(cell_t), pbc_mask,
                lo, hi, _ptr(ws), ws.numel(), _ptr(meta), _ptr(ent), n_central * row_cap, _ptr(status)))
        elif mode == "cell":
            if max_cells is None:
                max_cells = max(4096, 2 * n)
            ws = torch.empty(L.anihip_nbr_workspace_bytes(n, max_cells), dtype=torch.uint8, device=dev)
            _lib.check(L.anihip_nbr_build_cell(
                _stream(), C.byref(self.params), n, _ptr(species), _ptr(coords), _ptr(cell_t), pbc_mask, lo,
                hi, max_cells, _ptr(ws), ws.numel(), _ptr(meta), _ptr(ent), n_central * row_cap,
                _ptr(status)))
        else:
            raise ValueError(f"unknown neighbor mode {mode!r}")
        return NeighborRows(meta, ent, status, row_cap, lo, hi)

    def rows_from_half(self, species: Tensor, indices: Tensor, diff_vectors: Tensor, lo: int = 0,
                       hi: tp.Optional[int] = None, row_cap: int = 128) -> NeighborRows:
        """Neighbor rows from an external half list: indices [2, P] int64 (flattened atom indices),
        diff_vectors [P, 3] = r[indices[0]] - r[indices[1]] (+ image shift), neighbors.py:22-29,105-112."""
        _require_cuda(species, indices, diff_vectors)
        assert species.dtype == torch.int32 and species.is_contiguous()
        n = species.numel()
        hi = n if hi is None else hi
        dev = species.device
        idx = indices.to(torch.int64).contiguous()
        diff = diff_vectors.detach().to(torch.float32).contiguous()
        assert idx.dim() == 2 and idx.shape[0] == 2 and diff.shape == (idx.shape[1], 3)
        row_cap = int(min(max(row_cap, 1), _lib.MAX_RAD))
        n_central = max(hi - lo, 0)
        meta = torch.empty((n, _lib.META_WORDS), dtype=torch.int32, device=dev)
        ent = torch.empty((max(n_central, 1) * row_cap, 4), dtype=torch.float32, device=dev)
        status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        L = _lib.lib()
        ws = torch.empty(L.anihip_nbr_half_workspace_bytes(n_central), dtype=torch.uint8, device=dev)
        _lib.check(L.anihip_nbr_from_half(
            _stream(), C.byref(self.params), n, _ptr(species), idx.shape[1], _ptr(idx), _ptr(diff), lo, hi,
            _ptr(ws), ws.numel(), _ptr(meta), _ptr(ent), n_central * row_cap, _ptr(status)))
        return NeighborRows(meta, ent, status, row_cap, lo, hi)

    def rows_from_full(self, species: Tensor, coords: Tensor, ilist_unique: Tensor, jlist: Tensor,
                       numneigh: Tensor, row_cap: int = 128) -> NeighborRows:
        """Neighbor rows from a LAMMPS-style full list (aev/_computer.py:420-438): species [1, A] int32, coords
        [1, A, 3] float32 incl. ghost atoms; listed atom ilist_unique[g] has numneigh[g] consecutive jlist
        entries."""
        _require_cuda(species, coords, ilist_unique, jlist, numneigh)
        assert species.dtype == torch.int32 and coords.dtype == torch.float32
        assert species.is_contiguous() and coords.is_contiguous()
        n = species.numel()
        dev = coords.device
        il = ilist_unique.to(torch.int32).contiguous()
        jl = jlist.to(torch.int32).contiguous()
        nn = numneigh.to(torch.int32).contiguous()
        start = (torch.cumsum(nn.to(torch.int64), 0) - nn.to(torch.int64)).contiguous()
        row_cap = int(min(max(row_cap, 1), _lib.MAX_RAD))
        meta = torch.empty((n, _lib.META_WORDS), dtype=torch.int32, device=dev)
        ent = torch.empty((n * row_cap, 4), dtype=torch.float32, device=dev)
        status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().anihip_nbr_from_full(
            _stream(), C.byref(self.params), n, _ptr(species), _ptr(coords), il.numel(), _ptr(il), _ptr(nn),
            _ptr(start), _ptr(jl), _ptr(meta), _ptr(ent), n * row_cap, _ptr(status)))
        return NeighborRows(meta, ent, status, row_cap, 0, n, symmetric=False)

    def refresh_rows(self, species: Tensor, coords: Tensor, coords_build: Tensor, verlet: NeighborRows,
                     lo: int = 0, hi: tp.Optional[int] = None, row_cap: int = 128) -> NeighborRows:
        """Rows for the current coordinates from rows built at coords_build with cutoff Rcr + skin
        (anihip_nbr_refresh): no pair search, valid while no atom moved more than skin / 2."""
        _require_cuda(species, coords, coords_build)
        assert coords.dtype == torch.float32 and coords.is_contiguous() and coords_build.is_contiguous()
        n = species.numel()
        hi = n if hi is None else hi
        assert verlet.lo <= lo and hi <= verlet.hi and verlet.lo == 0, "Verlet rows must cover the central range"
        row_cap = int(min(max(row_cap, 1), _lib.MAX_RAD))
        n_central = max(hi - lo, 0)
        dev = coords.device
        meta = torch.empty((n, _lib.META_WORDS), dtype=torch.int32, device=dev)
        ent = torch.empty((max(n_central, 1) * row_cap, 4), dtype=torch.float32, device=dev)
        status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().anihip_nbr_refresh(
            _stream(), C.byref(self.params), n, lo, hi, _ptr(species), _ptr(coords), _ptr(coords_build),
            _ptr(verlet.meta), _ptr(verlet.ent), _ptr(meta), _ptr(ent), n_central * row_cap, _ptr(status)))
        return NeighborRows(meta, ent, status, row_cap, lo, hi)

    # ---- AEV ----------------------------------------------------------------------------------------
    @property
    def n_slabs(self) -> int:
        """Number of 32-wide slabs of an AEV row in slab order (include/anihip.h)."""
        if not self.tuned:
            return -(-self.L // 32)   # (a general grid: plain 32-column slabs of the row)
        S = self.params.num_species
        return (S + 1) // 2 + S * (S + 1) // 2

    def forward(self, species: Tensor, nbrs: NeighborRows, out: tp.Optional[Tensor] = None,
                slab_mask: tp.Optional[Tensor] = None, shard_rows: bool = False) -> Tensor:
        """AEV rows [N, L] for the central atoms nbrs.lo..nbrs.hi (other rows are left untouched).
        slab_mask (optional int32 [N], written for the same atoms): flags of the slabs of each row that are
        not identically zero, consumed by PackedNetworks.forward_backward.
        shard_rows=True: the result holds only the rows lo..hi ([hi - lo, L], row 0 = atom lo)."""
        _require_cuda(species, slab_mask)
        n = species.numel()
        rows0 = nbrs.lo if shard_rows else 0
        if out is None:
            if shard_rows:
                out = torch.empty((nbrs.hi - nbrs.lo, self.L), dtype=torch.float32, device=species.device)
            else:
                alloc = torch.empty if (nbrs.lo == 0 and nbrs.hi == n) else torch.zeros
                out = alloc((n, self.L), dtype=torch.float32, device=species.device)
        assert out.dtype == torch.float32 and out.is_contiguous()
        assert out.numel() == ((nbrs.hi - nbrs.lo) if shard_rows else n) * self.L
        if slab_mask is not None:
            assert slab_mask.dtype == torch.int32 and slab_mask.numel() == n and slab_mask.is_contiguous()
        _lib.check(_lib.lib().anihip_aev_forward(
            _stream(), C.byref(self.params), _ptr(self.table(species.device)), n, nbrs.lo, nbrs.hi,
            _ptr(species), _ptr(nbrs.meta), _ptr(nbrs.ent), _row_ptr(out, rows0, self.L), _ptr(slab_mask),
            _ptr(nbrs.status)))
        return out

    def forward_update(self, species: Tensor, nbrs: NeighborRows, shard_rows: bool = True) -> tp.Tuple[Tensor, Tensor]:
        """(AEV rows, slab flags) for the central atoms of nbrs in buffers the engine KEEPS between calls and updates in
        place (anihip_aev_forward_update): a row is zero outside its flagged slabs, and those zeros are written once -- a
        later call only touches the slabs that were or are flagged (0.6 KB instead of 4 KB per water atom and step).  The
        pair is valid until the next call of this method; one pair is kept (same number of atoms and central range, else a
        fresh one).  For the hot loop of energies_and_forces; ``forward`` hands out buffers of the caller's own."""
        _require_cuda(species)
        n = species.numel()
        rows = (nbrs.hi - nbrs.lo) if shard_rows else n
        # (the stream is part of the key: the kept pair is ordered by the stream that updates it, another stream gets its own)
        key = (n, nbrs.lo, nbrs.hi, rows, species.device, _stream())
        hit = self.__dict__.get("_rows_kept")
        if hit is None or hit[0] != key:
            self.__dict__["_rows_kept"] = None   # (release the old pair before the new one is allocated)
            # (two flag buffers, alternating: the kernel reads the previous call's and writes this call's)
            hit = [key, torch.zeros((rows, self.L), dtype=torch.float32, device=species.device),
                   torch.zeros(n, dtype=torch.int32, device=species.device),
                   torch.zeros(n, dtype=torch.int32, device=species.device)]
            self.__dict__["_rows_kept"] = hit
        _, out, prev, mask = hit
        _lib.check(_lib.lib().anihip_aev_forward_update(
            _stream(), C.byref(self.params), _ptr(self.table(species.device)), n, nbrs.lo, nbrs.hi,
            _ptr(species), _ptr(nbrs.meta), _ptr(nbrs.ent), _row_ptr(out, nbrs.lo if shard_rows else 0, self.L),
            _ptr(prev), _ptr(mask), _ptr(nbrs.status)))
        hit[2], hit[3] = mask, prev
        return out, mask

    def release_rows(self) -> None:
        """Free the buffers ``forward_update`` keeps (4 KB per central atom + two flag words per atom); the next call
        allocates and fills a fresh pair."""
        self.__dict__["_rows_kept"] = None
        self.__dict__["_rows_seen"] = None

    def rows_wanted(self, n: int, lo: int, hi: int, device: torch.device) -> bool:
        """Policy of the callers that MAY keep rows (models.ANI.energies_and_forces): a kept pair pays off from the second
        consecutive call with the same atom count and central range on -- a call whose sizes differ from the previous call's
        (batched screening, one-off evaluations) works on buffers of its own, pins nothing and memsets nothing."""
        key = (n, lo, hi, device, _stream())
        seen = self.__dict__.get("_rows_seen")
        self.__dict__["_rows_seen"] = key
        if seen != key:
            if self.__dict__.get("_rows_kept") is not None:
                self.__dict__["_rows_kept"] = None   # (sizes changed: the old pair is of no use any more)
            return False
        return True

    def __getstate__(self):
        # kept rows / flags are scratch tied to this object and its stream: never copied or pickled with it
        state = dict(self.__dict__)
        state.pop("_rows_kept", None)
        state.pop("_rows_seen", None)
        return state

    def __deepcopy__(self, memo):
        import copy

        new = object.__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def jvp(self, species: Tensor, nbrs: NeighborRows, tangent: Tensor) -> Tensor:
        """J t [N, L]: derivative of the AEV rows of nbrs' central atoms along the coordinate direction tangent [N, 3]
        (anihip_aev_jvp; the reference's cuaev double backward)."""
        _require_cuda(species, tangent)
        n = species.numel()
        t = tangent.detach().to(torch.float32).contiguous()
        assert t.numel() == 3 * n
        out = torch.empty((n, self.L), dtype=torch.float32, device=species.device) \
            if (nbrs.lo == 0 and nbrs.hi == n) else torch.zeros((n, self.L), dtype=torch.float32, device=species.device)
        _lib.check(_lib.lib().anihip_aev_jvp(
            _stream(), C.byref(self.params), _ptr(self.table(species.device)), n, nbrs.lo, nbrs.hi, _ptr(species),
            _ptr(nbrs.meta), _ptr(nbrs.ent), _ptr(t), _ptr(out), _ptr(nbrs.status)))
        return out

    def backward(self, species: Tensor, nbrs: NeighborRows, grad_aev: Tensor,
                 grad_coords: tp.Optional[Tensor] = None, shard_rows: bool = False,
                 virial: tp.Optional[Tensor] = None, slab_mask: tp.Optional[Tensor] = None,
                 fixed_point: bool = False) -> Tensor:
        """grad_coords [N,3] += d(sum grad_aev*aev)/d coords for the central atoms of nbrs.
        shard_rows=True: grad_aev holds only the rows lo..hi.
        slab_mask (int32 [N], as written by forward): grad_aev is valid only inside the flagged slabs of each row
        (what PackedNetworks.forward_backward leaves behind); None: whole rows are valid.
        virial (optional float64 [3,3], overwritten): sum_ij dE/d d_ij (x) d_ij over those central atoms.
        fixed_point=True: grad_coords is an int64 [N,3] accumulator in units of 2^-32 (ANIHIP_BWD_FIXED_POINT): the sums
        are order-independent, i.e. bit-reproducible; convert with ``fixed_to_float``."""
        _require_cuda(species, grad_aev, slab_mask)
        n = species.numel()
        assert grad_aev.dtype == torch.float32 and grad_aev.is_contiguous()
        assert grad_aev.numel() == ((nbrs.hi - nbrs.lo) if shard_rows else n) * self.L
        if slab_mask is not None:
            assert slab_mask.dtype == torch.int32 and slab_mask.numel() == n and slab_mask.is_contiguous()
        if grad_coords is None:
            grad_coords = torch.zeros((n, 3), dtype=torch.int64 if fixed_point else torch.float32, device=species.device)
        assert grad_coords.dtype == (torch.int64 if fixed_point else torch.float32) and grad_coords.is_contiguous()
        flags = (_lib.BWD_SYMMETRIC if nbrs.symmetric else 0) | (_lib.BWD_FIXED_POINT if fixed_point else 0)
        args = (_stream(), C.byref(self.params), _ptr(self.table(species.device)), n, nbrs.lo, nbrs.hi,
                _ptr(species), _ptr(nbrs.meta), _ptr(nbrs.ent),
                _row_ptr(grad_aev, nbrs.lo if shard_rows else 0, self.L), _ptr(slab_mask), flags,
                _ptr(grad_coords))
        if virial is None:
            _lib.check(_lib.lib().anihip_aev_backward(*args, _ptr(nbrs.status)))
        else:
            _require_cuda(virial)
            assert virial.dtype == torch.float64 and virial.is_contiguous() and virial.numel() == 9
            _lib.check(_lib.lib().anihip_aev_backward_virial(*args, _ptr(virial), _ptr(nbrs.status)))
        return grad_coords


FIXED_SCALE = 2.0 ** -32


def fixed_to_float(acc: Tensor) -> Tensor:
    """int64 fixed-point accumulators (units of 2^-32) -> float32."""
    return (acc.to(torch.float64) * FIXED_SCALE).to(torch.float32)


def _pad32(x: int) -> int:
    return (x + 31) // 32 * 32


class VerletRows:
    """The reference's VerletCellList (neighbors.py:759-884) as an API-COMPATIBILITY SHIM.  The reference reuses a pair list
    built with cutoff Rcr + skin until an atom has moved skin / 2 because its pair search dominates a step; this engine's
    O(N) cell-list search is 5 % of a step and a refresh of skin rows (read the longer rows back, update, screen, re-sort)
    measured SLOWER than or equal to the plain rebuild at every size from 3 000 to 2.3 M atoms (profiles/r06_md_bench.txt,
    profiles/r05_md_bench.txt: 0.47 / 0.41 ms at 3 k atoms, 0.66 / 0.65 at 12 k, 1.13 / 1.09 at 47 k, 1.83 / 1.74 at 98 k,
    3.55 / 3.46 at 192 k, 13.59 / 13.11 at 786 k).  So ``neighborlist="verlet_cell_list"`` is accepted, gives the same rows
    and results, and by default simply rebuilds every step (``rebuild_above = 0``); the reuse machinery (anihip_nbr_refresh,
    host-synchronising displacement check like the reference's) stays for callers who want the reference's semantics and
    is switched on with ``rebuild_above = float("inf")`` (tests/test_gpu_md.py, tools/md_bench.py --force-verlet)."""

    # systems of at least this many atoms are rebuilt every step (cell mode); 0 = always (see the class docstring)
    rebuild_above = 0

    def __init__(self, skin: float = 1.0) -> None:
        if skin <= 0.0:
            raise ValueError("skin must be a positive float")
        self.skin = float(skin)
        self.reset_cached_values()
        self.n_builds = 0
        self.n_reuses = 0
        self.n_direct = 0   # steps that took the plain pair search (rebuild_above)

    def reset_cached_values(self) -> None:
        self._wide: tp.Optional[AevEngine] = None
        self._rows: tp.Optional[NeighborRows] = None
        self._coords0: tp.Optional[Tensor] = None
        self._cell0: tp.Optional[Tensor] = None
        self._key: tp.Optional[tuple] = None

    def _can_use_prev_list(self, coords: Tensor, cell: tp.Optional[Tensor], key: tuple) -> bool:
        if self._rows is None or self._key != key or self._coords0.shape != coords.shape:
            return False
        if (cell is None) != (self._cell0 is None) or (cell is not None and not torch.equal(cell, self._cell0)):
            return False
        moved2 = (coords - self._coords0).pow(2).sum(-1).max()
        return bool(moved2 < (0.5 * self.skin) ** 2)

    def rows(self, eng: AevEngine, species: Tensor, coords: Tensor, cell: tp.Optional[Tensor], pbc, lo: int, hi: int,
             mode: str, row_cap: int) -> NeighborRows:
        if species.numel() >= self.rebuild_above and mode == "cell":
            self.n_direct += 1
            return eng.neighbors(species, coords, cell, pbc, lo=lo, hi=hi, mode=mode, row_cap=row_cap)
        cell_d = None if cell is None else cell.detach().to(device=coords.device, dtype=torch.float32)
        key = (eng.consts, tuple(species.shape), None if pbc is None else tuple(bool(b) for b in pbc), mode)
        if not self._can_use_prev_list(coords, cell_d, key):
            if self._wide is None or self._key is None or self._key[0] != eng.consts:
                # every stored neighbor in the "far" class: rows may then hold up to 256 entries (include/anihip.h)
                self._wide = AevEngine(eng.consts._replace(Rcr=eng.consts.Rcr + self.skin, Rca=1e-3))
            self._rows = self._wide.neighbors(species, coords, cell, pbc, lo=0, hi=species.numel(), mode=mode,
                                              row_cap=_lib.MAX_RAD)
            self._coords0 = coords.clone()
            self._cell0 = None if cell_d is None else cell_d.clone()
            self._key = key
            self.n_builds += 1
        else:
            self.n_reuses += 1
        out = eng.refresh_rows(species, coords, self._coords0, self._rows, lo, hi, row_cap)
        # a Verlet row that overflowed would silently lose neighbors: surface it in the refreshed rows' status
        out.status.bitwise_or_(self._rows.status & (_lib.ST_ROW_OVERFLOW | _lib.ST_ENTRY_OVERFLOW))
        return out


_N_CUS: tp.Dict[int, int] = {}


def _n_cus(dev: torch.device) -> int:
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    if i not in _N_CUS:
        _N_CUS[i] = int(torch.cuda.get_device_properties(i).multi_processor_count)
    return _N_CUS[i]


def species_column_map(num_species: int, radial_len: int, aev_len: int, order: tp.Sequence[int]) -> np.ndarray:
    """AEV column of the reference layout (aev/_computer.py: radial blocks by species, angular blocks by the triu index of
    the species pair) for every column of the SAME layout written with the species relabelled ``new = order.index(old)``:
    ``aev_relabelled[:, c] == aev_reference[:, map[c]]``.  Used to permute the input rows of the layer-0 weights when the
    engine numbers the species of a system "present ones first" (models.ANI.compact_species)."""
    S = num_species
    nr, pairs = radial_len // S, S * (S + 1) // 2
    nb = (aev_len - radial_len) // pairs
    assert nr * S == radial_len and radial_len + nb * pairs == aev_len and sorted(order) == list(range(S))

    def triu(a: int, b: int) -> int:   # index of the pair (a <= b) in row-major upper-triangular order
        return a * S - (a * (a - 1)) // 2 + (b - a)

    out = np.empty(aev_len, dtype=np.int64)
    for s_new in range(S):
        out[s_new * nr:(s_new + 1) * nr] = order[s_new] * nr + np.arange(nr)
    for a in range(S):
        for b in range(a, S):
            oa, ob = sorted((order[a], order[b]))
            out[radial_len + triu(a, b) * nb:radial_len + (triu(a, b) + 1) * nb] = radial_len + triu(oa, ob) * nb + np.arange(nb)
    return out


class PackedNetworks:
    """Ensemble parameters in the MFMA-friendly layout of include/anihip.h (members concatenated, widths
    padded to 32, transposed copies for the backward GEMMs); cf. BmmAtomicNetwork, nn/_infer.py:141-161."""

    # anihip_mlp_desc.flags (_lib.MLP_FLAG_*) of forward_backward: ``flags`` of an instance, else this class default
    # (0 = the library chooses from the problem size).  The library itself reads no environment variables.
    default_flags: int = 0
    flags: tp.Optional[int] = None
    pinned: int = 0   # live HIP graphs that captured this object's buffers (GraphedEnergiesForces)

    def __init__(self, weights: tp.Sequence[tp.Sequence[tp.Sequence[Tensor]]],
                 biases: tp.Sequence[tp.Sequence[tp.Sequence[Tensor]]], aev_len: int, celu_alpha: float,
                 device: torch.device, precision: str = "f16x3", radial_len: tp.Optional[int] = None,
                 activation: str = "celu") -> None:
        """weights[m][s][l]: [out, in] (torch.nn.Linear layout) of member m, species s, layer l.

        radial_len: length R of the radial part of the AEV; the layer-0 fp16 planes are then stored in slab
        order (include/anihip.h) so that forward_backward can skip all-zero AEV blocks given slab masks.
        Default: 16 S when aev_len has the ANI form 16 S + 32 S(S+1)/2, else plain order.

        precision: "fp32" (exact fp32 MFMA) or "f16x3" (split-fp16 three-product MFMA, ~4e-7 relative
        error per product, see include/anihip.h)."""
        if precision not in ("fp32", "f16x3"):
            raise ValueError(f"unknown MLP precision {precision!r}")
        if activation not in ("celu", "gelu"):
            raise ValueError(f"unknown activation {activation!r}: the network kernels have celu and gelu")
        # (GELU: inference runs through the fused f16x3 kernel only -- anihip_mlp_forward_backward refuses an fp32 GELU pack --;
        # an fp32 GELU pack serves the TRAINING passes, which keep the pre-activations: train_forward, weight_grads,
        # tangent_weight_grads)
        self.precision = precision
        self.activation = activation
        M, S = len(weights), len(weights[0])
        nl = len(weights[0][0])
        if not (2 <= nl <= _lib.MAX_LAYERS):
            raise ValueError(f"networks must have 2..{_lib.MAX_LAYERS} Linear layers")
        if aev_len % 16 != 0:
            raise ValueError("AEV length must be a multiple of 16")
        self.M, self.S, self.nl, self.aev_len = M, S, nl, aev_len
        self.device = device
        self.shapes = [[tuple(weights[0][s][l].shape) for l in range(nl)] for s in range(S)]   # unpadded [out, in]
        # The layouts are produced by the library itself (anihip_mlp_pack, csrc/pack.hip): the same call a caller without
        # this package makes.  Parameters are handed over as contiguous fp32 tensors where they live (device or host).
        sh = _lib.MlpShape()
        sh.n_members, sh.num_species, sh.n_layers, sh.aev_len = M, S, nl, aev_len
        sh.aev_radial_len = -1 if radial_len is None else int(radial_len)
        sh.precision = _lib.MLP_F16X3 if precision == "f16x3" else _lib.MLP_FP32
        sh.activation = _lib.ACT_GELU if activation == "gelu" else _lib.ACT_CELU
        sh.celu_alpha = celu_alpha
        for s in range(S):
            if weights[0][s][nl - 1].shape[0] != 1:
                raise ValueError("final layer must have one output")
            for l in range(nl):
                sh.out_dims[s][l] = int(weights[0][s][l].shape[0])
        src, on_dev = [], None
        wp, bp = (C.c_void_p * (M * S * nl))(), (C.c_void_p * (M * S * nl))()
        for m in range(M):
            for s in range(S):
                for l in range(nl):
                    W = weights[m][s][l].detach().to(torch.float32).contiguous()
                    b = biases[m][s][l].detach().to(torch.float32).contiguous()
                    if tuple(W.shape) != self.shapes[s][l] or b.numel() != W.shape[0]:
                        raise ValueError("every member must have the same layer shapes")
                    if on_dev is None:
                        on_dev = W.is_cuda
                    if W.is_cuda != on_dev or b.is_cuda != on_dev:
                        raise ValueError("parameters must all live on the host or all on a device")
                    src += [W, b]
                    wp[(m * S + s) * nl + l], bp[(m * S + s) * nl + l] = W.data_ptr(), b.data_ptr()
        L = _lib.lib()
        need = L.anihip_mlp_pack_bytes(C.byref(sh))
        if need == 0:
            raise ValueError("libanihip: " + L.anihip_last_error().decode())
        self._buf = torch.empty(need, dtype=torch.uint8, device=device)
        d = _lib.MlpDesc()
        if on_dev:
            torch.cuda.current_stream(src[0].device).synchronize()   # (the library reads the parameters with blocking copies)
        stream = _stream() if self._buf.is_cuda else None
        _lib.check(L.anihip_mlp_pack(stream, C.byref(sh), wp, bp, 1 if on_dev else 0, _ptr(self._buf), need,
                                     1 if self._buf.is_cuda else 0, C.byref(d)))
        del src
        self.radial_len = d.aev_radial_len
        self.desc = d
        self._ws: tp.Optional[Tensor] = None
        self._train_ws: tp.Optional[Tensor] = None

    def array(self, ptr: int, shape: tp.Sequence[int], dtype: torch.dtype = torch.float32) -> Tensor:
        """View of one packed array (a pointer of ``desc``) inside the buffer anihip_mlp_pack filled."""
        off = int(ptr) - self._buf.data_ptr()
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        assert 0 <= off and off + n <= self._buf.numel(), "pointer outside the packed buffer"
        return self._buf[off:off + n].view(dtype).view(*shape)

    def workspace(self, n_central: int, want_grad: bool = True) -> Tensor:
        """Scratch of one anihip_mlp_forward_backward call over n_central atoms: what the kernels that call runs touch
        (with the layer-0 backward inside the fused kernel ~100 B per atom: index lists, tile table, per-member energies;
        19.5 KB per atom for the layer-by-layer kernels of an ANI-2x pack)."""
        need = _lib.lib().anihip_mlp_forward_backward_workspace_bytes(C.byref(self.desc), n_central, int(want_grad))
        if self._ws is None or self._ws.numel() < need:
            if self.pinned and self._ws is not None:
                # a captured HIP graph replays into self._ws: never free or replace it; a larger eager call gets a
                # buffer of its own
                return torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward_backward(self, species: Tensor, aev: Tensor, lo: int = 0, hi: tp.Optional[int] = None,
                         want_grad: bool = True, want_members: bool = False, chunk: tp.Optional[int] = None,
                         atomic_e: tp.Optional[Tensor] = None, grad_aev: tp.Optional[Tensor] = None,
                         slab_mask: tp.Optional[Tensor] = None, shard_rows: bool = False, tile_hint: int = 0,
                         plain_slabs: bool = False) -> tp.Tuple[Tensor, tp.Optional[Tensor], tp.Optional[Tensor]]:
        """Per-atom ensemble-mean energies [N], d e/d aev [N,L] (optional), member energies [M,N].

        slab_mask (int32 [N] from AevEngine.forward): the layer-0 GEMMs skip AEV slabs no atom of a row tile
        flags; grad_aev is then only defined inside the flagged slabs (all AevEngine.backward reads).  The flags must
        be in the order of this pack's layer-0 planes: the ANI slab order of the 16 / 32-column grids (radial_len > 0),
        or -- plain_slabs=True, general grids, radial_len == 0 -- the plain 32-column slabs of the row.
        shard_rows=True: aev (and the returned / given grad_aev) hold only the rows lo..hi.
        tile_hint: _lib.MLP_FLAG_SMALL_TILES / MLP_FLAG_BIG_TILES from a caller that knows the composition (used unless the
        instance / class flags already choose a layer-0 tiling)."""
        _require_cuda(species, aev, slab_mask)
        if slab_mask is not None:
            assert slab_mask.dtype == torch.int32 and slab_mask.numel() == species.numel()
            if (self.radial_len == 0) != plain_slabs or (plain_slabs and (self.aev_len > 1024 or self.precision != "f16x3")):
                slab_mask = None   # (flags in another order than the planes': multiply every slab)
        n = species.numel()
        hi = n if hi is None else hi
        rows = (hi - lo) if shard_rows else n
        rows0 = lo if shard_rows else 0
        assert aev.dtype == torch.float32 and aev.is_contiguous() and aev.numel() == rows * self.aev_len
        dev = aev.device
        full = (lo == 0 and hi == n) or shard_rows
        if atomic_e is None:
            atomic_e = (torch.empty if (lo == 0 and hi == n) else torch.zeros)(n, dtype=torch.float32, device=dev)
        if want_grad and grad_aev is None:
            grad_aev = (torch.empty if full else torch.zeros)((rows, self.aev_len), dtype=torch.float32, device=dev)
        if want_grad:
            assert grad_aev.dtype == torch.float32 and grad_aev.is_contiguous()
            assert grad_aev.numel() == rows * self.aev_len
        member_e = torch.zeros((self.M, n), dtype=torch.float32, device=dev) if want_members else None
        L = _lib.lib()
        # equal chunks (a short last chunk would leave most CUs idle in its tail), up to twice the preferred size when
        # that saves rounds of the 256-row layer-0 backward GEMM: its one-workgroup-per-CU tiles run in whole rounds
        # over the CUs (a rank's 292 k-atom shard of the 2.3 M-atom box: 2 chunks x 3 rounds -> 1 chunk x 5 rounds)
        nn_ = hi - lo
        n_cu = _n_cus(dev)
        fl = PackedNetworks.default_flags if self.flags is None else self.flags
        tiles = _lib.MLP_FLAG_SMALL_TILES | _lib.MLP_FLAG_BIG_TILES
        if not fl & tiles:
            fl |= tile_hint & tiles
        fl |= tile_hint & ~tiles   # (other switches a caller hands over with the hint, e.g. MLP_FLAG_BWD_TWO_PRODUCTS)
        self.desc.flags = fl   # (before the workspace queries: the flags choose the kernels)
        if chunk is None:
            # ONE call when its scratch is small -- the fused kernel with the layer-0 backward inside keeps everything but
            # ~100 B per atom in LDS: one persistent launch and one tail instead of one per 2^20 atoms (-0.5 ms per step at
            # 2.34 M atoms) -- else launch groups of 2^20 atoms (19.5 KB of activations per atom for an ANI-2x pack)
            chunk = 1 << 20
            if nn_ > chunk and L.anihip_mlp_forward_backward_workspace_bytes(C.byref(self.desc), nn_, int(want_grad)) <= 512 * nn_:
                chunk = nn_

        def plan(nc: int) -> tp.Tuple[int, int]:
            st_ = max(1, -(-(-(-nn_ // nc)) // 256) * 256)
            return nc * -(-(st_ // 256 + self.S) // n_cu), st_

        nc_hi = max(1, -(-nn_ // chunk))
        cands = range(max(1, -(-nn_ // (2 * chunk))), nc_hi + 3) if want_grad and nn_ >= 65536 else (nc_hi,)
        nchunk = min(cands, key=lambda nc: (plan(nc)[0], nc))
        step = plan(nchunk)[1]
        for c0 in range(lo, hi, step):
            c1 = min(hi, c0 + step)
            ws = self.workspace(c1 - c0, want_grad)
            _lib.check(L.anihip_mlp_forward_backward(
                _stream(), C.byref(self.desc), n, c0, c1, _ptr(species), _row_ptr(aev, rows0, self.aev_len),
                _ptr(slab_mask), _ptr(ws), ws.numel(), _ptr(atomic_e),
                _row_ptr(grad_aev, rows0, self.aev_len) if want_grad else None, _ptr(member_e)))
        return atomic_e, (grad_aev if want_grad else None), member_e

    def refresh(self, weights, biases, fused_only: bool = False) -> None:
        """Re-read the parameter values (same shapes as at construction) into the packed arrays on the device
        (anihip_mlp_repack): w / wt / bias of an fp32 pack, every layout -- fp32 arrays, {hi, lo} fp16 planes, fragment
        orders, fused_bounds -- of an f16x3 pack (with the weight scales it was built with; ``scale_overflowed()`` tells,
        one call late, whether a weight has outgrown them)."""
        M, S, nl = self.M, self.S, self.nl
        ptrs = [t.data_ptr() for m in range(M) for s in range(S) for l in range(nl)
                for t in (weights[m][s][l], biases[m][s][l])]
        key = tuple(ptrs)
        if getattr(self, "_src_key", None) != key:
            for m in range(M):
                for s in range(S):
                    for l in range(nl):
                        W, b = weights[m][s][l], biases[m][s][l]
                        if (tuple(W.shape) != self.shapes[s][l] or W.dtype != torch.float32 or not W.is_contiguous()
                                or b.dtype != torch.float32 or not b.is_contiguous() or W.device != self.device):
                            raise ValueError("refresh() needs contiguous fp32 parameters of the packed shapes")
            self._src_tab = torch.tensor(ptrs, dtype=torch.int64, device=self.device)
            self._src_key = key
            self._out_in = np.asarray(self.shapes, dtype=np.int32).reshape(-1).copy()
        status = None
        if self.precision == "f16x3":
            if getattr(self, "_repack_status", None) is None:
                self._repack_status = torch.zeros(1, dtype=torch.int32, device=self.device)
                self._repack_poll = None
            status = self._repack_status
        # fused_only (f16x3): what the fused kernel and the fast training pass read; the layer-by-layer layouts go stale until
        # a full refresh (self.stale_layouts)
        flags = _lib.REPACK_FUSED_ONLY if (fused_only and self.precision == "f16x3") else 0
        self.stale_layouts = bool(flags)
        _lib.check(_lib.lib().anihip_mlp_repack(_stream(), C.byref(self.desc), _ptr(self._src_tab),
                                                self._out_in.ctypes.data, _ptr(status), flags))
        if status is not None and not torch.cuda.is_current_stream_capturing() and self._repack_poll is None:
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(status, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._repack_poll = (host, ev)

    def scale_overflowed(self) -> bool:
        """Did a refresh find a weight outside the fp16 range of its layer's scale?  Reads what the PREVIOUS refresh queued
        (no wait unless the host runs a whole step ahead); the caller then packs again on the host (new scales)."""
        poll = getattr(self, "_repack_poll", None)
        if poll is None or torch.cuda.is_current_stream_capturing():
            return False
        host, ev = poll
        if not ev.query():
            return False
        self._repack_poll = None
        return bool(host[0])

    def fast_training(self) -> bool:
        """Do the training passes of this pack run through the fused network kernel (anihip.h: ANIHIP_MLP_F16X3, CELU, three
        hidden layers <= 256 wide)?"""
        return (self.precision == "f16x3" and self.activation == "celu" and self.nl == 4
                and all(self.desc.net[s].dims[l] <= 256 for s in range(self.S) for l in (1, 2, 3))
                and self.aev_len <= 1024)

    def _freshen_layouts(self, fused_route: bool) -> None:
        """A refresh(..., fused_only=True) leaves the layer-by-layer layouts (w / wt / wh / wth) at the parameters of the pack
        before it.  A call that the library serves layer by layer -- anything but the fused training pass without d Loss /
        d aev -- must not read them: run the full repack from the parameter table of the last refresh first (round-5
        advice: the C side decides the route from the descriptor on its own, so the guard sits where both are known)."""
        if getattr(self, "stale_layouts", False) and not fused_route:
            _lib.check(_lib.lib().anihip_mlp_repack(_stream(), C.byref(self.desc), _ptr(self._src_tab),
                                                    self._out_in.ctypes.data, _ptr(getattr(self, "_repack_status", None)), 0))
            self.stale_layouts = False

    def train_forward(self, species: Tensor, aev: Tensor) -> tp.Tuple[Tensor, Tensor]:
        """First half of a training step: exact-fp32 forward that keeps the activations.  Returns (atomic_e [N],
        workspace) -- hand the workspace to weight_grads(..., workspace=ws) for the backward half."""
        _require_cuda(species, aev)
        self._freshen_layouts(self.fast_training())
        n = species.numel()
        assert aev.dtype == torch.float32 and aev.is_contiguous() and aev.numel() == n * self.aev_len
        L = _lib.lib()
        ws = torch.empty(L.anihip_mlp_train_workspace_bytes(C.byref(self.desc), n), dtype=torch.uint8,
                         device=aev.device)
        atomic_e = torch.zeros(n, dtype=torch.float32, device=aev.device)
        _lib.check(L.anihip_mlp_train_forward(_stream(), C.byref(self.desc), n, 0, n, _ptr(species), _ptr(aev),
                                              _ptr(ws), ws.numel(), _ptr(atomic_e)))
        return atomic_e, ws

    def _grad_buffers(self, dev):
        """One flat fp32 buffer holding every gradient array of anihip_species_grads + the ctypes pointer table."""
        M, S, nl, d = self.M, self.S, self.nl, self.desc
        sizes = []
        for s in range(S):
            dims = [d.net[s].dims[l] for l in range(nl + 1)]
            for l in range(nl):
                sizes += [M * dims[l] * dims[l + 1], M * dims[l + 1]]
        offs = np.concatenate([[0], np.cumsum([(x + 63) // 64 * 64 for x in sizes])])
        buf = torch.empty(int(offs[-1]), dtype=torch.float32, device=dev)
        sg = (_lib.SpeciesGrads * S)()
        q = 0
        for s in range(S):
            for l in range(nl):
                sg[s].gw[l] = buf.data_ptr() + 4 * int(offs[q])
                sg[s].gbias[l] = buf.data_ptr() + 4 * int(offs[q + 1])
                q += 2
        return buf, sg, sizes, offs

    def _unpack_grads(self, total: Tensor, sizes, offs):
        """Flat gradient buffer -> gw[m][s][l] [out, in], gb[m][s][l] [out] (views when the widths are unpadded)."""
        M, S, nl, d = self.M, self.S, self.nl, self.desc
        gw = [[[None] * nl for _ in range(S)] for _ in range(M)]
        gb = [[[None] * nl for _ in range(S)] for _ in range(M)]
        q = 0
        for s in range(S):
            dims = [d.net[s].dims[l] for l in range(nl + 1)]
            for l in range(nl):
                out, inn = self.shapes[s][l]
                w = total[int(offs[q]): int(offs[q]) + sizes[q]]
                b = total[int(offs[q + 1]): int(offs[q + 1]) + sizes[q + 1]]
                q += 2
                if l == nl - 1:
                    wv, bv = w.view(M, 1, dims[l]), b.view(M, 1)
                else:
                    wv, bv = w.view(M, dims[l + 1], dims[l]), b.view(M, dims[l + 1])   # nn.Linear layout
                for m in range(M):
                    gw[m][s][l] = wv[m, :out, :inn]
                    gb[m][s][l] = bv[m, :out]
        return gw, gb

    def tangent_weight_grads(self, species: Tensor, aev: Tensor, tangent: Tensor):
        """Second-order pass of force training (anihip_mlp_tangent_weight_grads): gradients with respect to every
        weight and bias of  S = sum_i tangent_i . d atomic_e[i] / d aev_i,  plus the per-atom terms of S [N]."""
        _require_cuda(species, aev, tangent)
        n = species.numel()
        dev = aev.device
        assert aev.dtype == torch.float32 and aev.is_contiguous() and aev.numel() == n * self.aev_len
        t = tangent.to(torch.float32).contiguous()
        assert t.numel() == n * self.aev_len
        buf, sg, sizes, offs = self._grad_buffers(dev)
        L = _lib.lib()
        ws = torch.empty(L.anihip_mlp_tangent_workspace_bytes(C.byref(self.desc), n), dtype=torch.uint8, device=dev)
        de = torch.zeros(n, dtype=torch.float32, device=dev)
        _lib.check(L.anihip_mlp_tangent_weight_grads(
            _stream(), C.byref(self.desc), n, 0, n, _ptr(species), _ptr(aev), _ptr(t), _ptr(ws), ws.numel(), sg,
            _ptr(de)))
        gw, gb = self._unpack_grads(buf, sizes, offs)
        return gw, gb, de

    def flat_grad_target(self, w_ptr, b_ptr, member_stride: int):
        """anihip_species_grads table for gradients that go straight into a flat buffer (torchani_amd.optim.Adam):
        w_ptr[s][l] / b_ptr[s][l] = device addresses of MEMBER 0's weight / bias gradient, member m's lie member_stride floats
        further on; accumulated into (the optimizer zeroes the buffer).  Needs unpadded widths."""
        sg = (_lib.SpeciesGrads * self.S)()
        for s in range(self.S):
            for l in range(self.nl):
                out, inn = self.shapes[s][l]
                if self.desc.net[s].dims[l] != inn or (l < self.nl - 1 and self.desc.net[s].dims[l + 1] != out):
                    raise ValueError("flat gradient targets need layer widths that are multiples of 32")
                sg[s].gw[l], sg[s].gbias[l] = int(w_ptr[s][l]), int(b_ptr[s][l])
            sg[s].member_stride, sg[s].accumulate = int(member_stride), 1
        return sg

    def weight_grads(self, species: Tensor, aev: Tensor, grad_atomic_e: Tensor,
                     want_grad_aev: bool = False, chunk: int = 1 << 16, workspace: tp.Optional[Tensor] = None,
                     target=None):
        """Training pass (anihip_mlp_weight_grads): gradients of  sum_i grad_atomic_e[i] * atomic_e[i]  with respect
        to every weight and bias, returned in torch.nn.Linear layout: gw[m][s][l] [out, in], gb[m][s][l] [out];
        plus atomic_e [N] and, optionally, d Loss / d aev [N, L].  Replaces torch autograd through
        nn/_core.py:146-149 / nn/_containers.py:377-421,608-636."""
        _require_cuda(species, aev, grad_atomic_e)
        self._freshen_layouts(self.fast_training() and not want_grad_aev)
        n = species.numel()
        dev = aev.device
        assert aev.dtype == torch.float32 and aev.is_contiguous() and aev.numel() == n * self.aev_len
        g_at = grad_atomic_e.to(torch.float32).contiguous().view(-1)
        assert g_at.numel() == n
        d = self.desc
        atomic_e = torch.zeros(n, dtype=torch.float32, device=dev)
        grad_aev = torch.zeros((n, self.aev_len), dtype=torch.float32, device=dev) if want_grad_aev else None
        L = _lib.lib()
        total = None
        if workspace is not None:
            chunk = max(n, 1)   # the forward half ran over all atoms at once (train_forward)
        for c0 in range(0, max(n, 1), chunk):
            c1 = min(n, c0 + chunk)
            if target is not None:   # (flat_grad_target: accumulated in place, nothing to unpack)
                buf, sg, sizes, offs = None, target, None, None
            else:
                buf, sg, sizes, offs = self._grad_buffers(dev)
            need = L.anihip_mlp_train_workspace_bytes(C.byref(d), c1 - c0)
            if workspace is not None:
                ws = workspace
                assert ws.numel() >= need
            else:
                if self._train_ws is None or self._train_ws.numel() < need:
                    self._train_ws = torch.empty(need, dtype=torch.uint8, device=dev)
                ws = self._train_ws
            _lib.check(L.anihip_mlp_weight_grads(
                _stream(), C.byref(d), n, c0, c1, _ptr(species), _ptr(aev), _ptr(g_at), _ptr(ws),
                ws.numel(), sg, _ptr(atomic_e), _ptr(grad_aev), 1 if workspace is not None else 0))
            if buf is not None:
                total = buf if total is None else total.add_(buf)
        if target is not None:
            return None, None, atomic_e, grad_aev
        gw, gb = self._unpack_grads(total, sizes, offs)
        return gw, gb, atomic_e, grad_aev


def energy_reduce(species: Tensor, atomic_e: Tensor, sae: tp.Optional[Tensor], lo: int = 0,
                  hi: tp.Optional[int] = None) -> Tensor:
    """Molecular energies [C] in float64 = sum over atoms lo..hi of (atomic_e + sae[species])."""
    _require_cuda(species, atomic_e, sae)
    Cn, A = species.shape
    n = Cn * A
    hi = n if hi is None else hi
    out = torch.empty(Cn, dtype=torch.float64, device=species.device)
    if sae is not None:
        assert sae.dtype == torch.float64
    _lib.check(_lib.lib().anihip_energy_reduce(
        _stream(), Cn, A, lo, hi, _ptr(species), _ptr(atomic_e), _ptr(sae), _ptr(out)))
    return out


def energy_forces_finish(species: Tensor, atomic_e: Tensor, sae: tp.Optional[Tensor], grad_coords: Tensor, lo: int = 0,
                         hi: tp.Optional[int] = None) -> Tensor:
    """energy_reduce, and ``grad_coords`` (float32, contiguous) negated IN PLACE into forces, in one launch for batches of
    small molecules (anihip_energy_forces_finish).  Returns the molecular energies [C] in float64."""
    _require_cuda(species, atomic_e, sae, grad_coords)
    Cn, A = species.shape
    hi = Cn * A if hi is None else hi
    assert grad_coords.dtype == torch.float32 and grad_coords.is_contiguous()
    if sae is not None:
        assert sae.dtype == torch.float64
    out = torch.empty(Cn, dtype=torch.float64, device=species.device)
    _lib.check(_lib.lib().anihip_energy_forces_finish(
        _stream(), Cn, A, lo, hi, _ptr(species), _ptr(atomic_e), _ptr(sae), _ptr(out), _ptr(grad_coords),
        grad_coords.numel()))
    return out
