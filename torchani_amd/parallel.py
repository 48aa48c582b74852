"""Data-parallel sharding of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no multi-device code at all; the decomposition is ours:
  * central atoms (flattened molecule-major) are split into contiguous, equal ranges, one per rank --
    for batches of molecules this is a split over molecules, for one big periodic box a split over the
    atoms of the box; every rank keeps the full coordinate array (37 MB for 2.3 M atoms), so building the
    shard's neighbor rows needs no halo exchange;
  * ONE collective per step: a central atom pushes gradient onto its neighbors, which may belong to other ranks,
    so a shared system ends with an all-reduce of the [N,3] fp32 force array (27.6 MB at 2.3 M atoms; a few
    hundred microseconds over xGMI against milliseconds of compute), and the fp64 partial energies (and the
    virial) ride in the SAME buffer as four exactly-summable fp32 parts each (split_exact / join_exact): the
    sum over ranks is exact and does not depend on the reduction order.  Batches whose molecules do not
    straddle ranks reduce only that 16-byte-per-molecule tail (reduce_forces=False);
  * training (BASELINE config 5): minibatches are split over molecules, every rank back-propagates its share with
    the engine's training pass and the weight gradients (13.7 M floats for ANI-2x x 8) are summed with ONE bucketed
    all-reduce of a flat buffer (all_reduce_gradients) -- 55 MB, per-link bound on xGMI like the force all-reduce.
"""
from __future__ import annotations

import math
import os
import typing as tp

import torch


def shard_bounds(n: int, world: int) -> tp.List[int]:
    """world+1 monotone boundaries splitting range(n) into near-equal contiguous shards."""
    base, rem = divmod(n, world)
    out = [0]
    for r in range(world):
        out.append(out[-1] + base + (1 if r < rem else 0))
    return out


def shard_range(n: int, group=None, rank: tp.Optional[int] = None, world: tp.Optional[int] = None
                ) -> tp.Tuple[int, int]:
    """[lo, hi) of this rank's central atoms."""
    if world is None:
        if group is None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return 0, n
        if group is None:
            return 0, n  # sharding is opt-in: pass the process group explicitly
        world = torch.distributed.get_world_size(group)
        rank = torch.distributed.get_rank(group)
    b = shard_bounds(n, world)
    return b[rank], b[rank + 1]


# An fp64 number x, |x| < 2^31, is written as four fp32 parts on fixed grids 2^12, 2^-8, 2^-28, 2^-48, each part an integer
# multiple of its grid below 2^19 in magnitude: sums of up to 32 such parts stay below 2^24 grid units, i.e. fp32
# addition of them is EXACT in any order, and the four sums give x to 2^-49 (1.8e-15) -- how fp64 scalars travel
# through an fp32 all-reduce.
_GRIDS = (12, -8, -28, -48)
EXACT_MAX_WORLD = 32


def split_exact(x: torch.Tensor) -> torch.Tensor:
    """float64 [...] -> float32 [..., 4] exactly-summable parts (see above)."""
    r = x.to(torch.float64)
    parts = []
    for k in _GRIDS:
        g = 2.0 ** k
        p = torch.round(r / g) * g
        parts.append(p)
        r = r - p
    return torch.stack(parts, dim=-1).to(torch.float32)


def join_exact(parts: torch.Tensor) -> torch.Tensor:
    """float32 [..., 4] (summed over ranks) -> float64 [...]."""
    return parts.to(torch.float64).sum(dim=-1)


# development aid: TORCHANI_AMD_FORCE_GROUP=1 creates the process group and runs every collective of a step even at world
# size 1 -- the only way to execute the RCCL calls themselves on a box with a single GPU (tools/gpu_rccl_world1.sh)
FORCE_COLLECTIVES = os.environ.get("TORCHANI_AMD_FORCE_GROUP") == "1"


def init_from_env(backend: tp.Optional[str] = None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world, local_rank, group-or-None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not FORCE_COLLECTIVES:
        return rank, world, local, None
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.distributed.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            torch.distributed.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
    return rank, world, local, torch.distributed.group.WORLD


def all_reduce_gradients(parameters: tp.Iterable[torch.nn.Parameter], group=None, average: bool = False) -> None:
    """Sum (or average) the .grad of the given parameters over the ranks of ``group`` with one collective: the
    gradients are packed into a single flat bucket (one large all-reduce instead of hundreds of small ones; parameters
    without a gradient contribute zeros so every rank sends the same layout), reduced and unpacked in place."""
    if group is None or torch.distributed.get_world_size(group) == 1:
        return
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return
    flat_group = _one_flat_group(params)
    if flat_group is not None:
        # torchani_amd.optim.Adam keeps the gradients of its parameters as views of ONE buffer: that buffer is the bucket
        stage = flat_group.grad.is_cuda and torch.distributed.get_backend(group) == "gloo"
        buf = flat_group.grad.cpu() if stage else flat_group.grad
        torch.distributed.all_reduce(buf, group=group)
        if average:
            buf /= torch.distributed.get_world_size(group)
        if stage:
            flat_group.grad.copy_(buf)
        return
    dev, dtype = params[0].device, params[0].dtype
    flat = torch.zeros(sum(p.numel() for p in params), dtype=dtype, device=dev)
    off = 0
    for p in params:
        if p.grad is not None:
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        off += p.numel()
    torch.distributed.all_reduce(flat, group=group)
    if average:
        flat /= torch.distributed.get_world_size(group)
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()


def _one_flat_group(params: tp.List[torch.nn.Parameter]):
    """The flat parameter group of torchani_amd.optim.Adam that holds exactly these parameters, every gradient still its view
    of the group's buffer -- or None."""
    tag = getattr(params[0], "_anihip_flat", None)
    f = tag[0]() if tag is not None else None
    if f is None or len(f.params) != len(params):
        return None
    for i, p in enumerate(params):
        if f.params[i] is not p or p.grad is not f.grad_views[i]:
            return None
    return f


# ---- spatial shards of ONE big system (slabs of the cell-sorted order + a cutoff-wide halo) -------------------
class SpatialShards:
    """Slab decomposition of one large system for ``world`` ranks, independent of the order of the input atoms.

    Atoms are sorted by their fractional coordinate along one cell axis (the longest; a bounding-box axis without PBC) and
    the sorted order is cut into ``world`` contiguous ranges of equal atom count: rank r OWNS the central atoms of range r,
    a slab of the box.  Its neighbor rows need the atoms within ``cutoff`` of the slab as well: the ``n_left`` sorted
    positions before the range and the ``n_right`` behind it (circularly under PBC).  Every rank therefore works on a LOCAL
    system [left halo | owned | right halo] -- binning, neighbor rows, AEVs, networks and the force buffer are all sized by
    the slab, nothing is replicated -- and the only forces it cannot finish alone are the pushes of its own central atoms
    onto halo atoms, which belong to the neighboring slabs.

    One collective per step: every rank contributes its halo rows (+ its partial energies / virial, fp64 carried as pairs
    of fp32 words) to ONE all-gather; ``exchange`` adds what the other ranks hold for this rank's atoms.  For the
    2.34 M-atom water box on 8 ranks that is 2 x 5.1 A of a 35.5 A slab: ~1 MB per rank instead of the 27.6 MB all-reduce
    of a replicated force array.

    The plan (who holds halo rows of whom) is computed identically on every rank from the sorted coordinate array and
    2 * world boundary searches (one small device-to-host copy when the partition is built)."""

    def __init__(self, coords: torch.Tensor, cell: tp.Optional[torch.Tensor], pbc: tp.Optional[tp.Sequence[bool]],
                 world: int, rank: int, cutoff: float, species: tp.Optional[torch.Tensor] = None, skin: float = 0.0) -> None:
        """skin > 0: the halos are cut ``cutoff + skin`` wide, so the partition stays valid while no atom has moved more than
        skin / 2 since it was built (``still_valid``; an MD driver rebuilds it every few dozen steps instead of every step)."""
        x = coords.detach().reshape(-1, 3).to(torch.float32)
        n = x.shape[0]
        dev = x.device
        self.n, self.world, self.rank, self.cutoff, self.skin = n, world, rank, float(cutoff), float(skin)
        reach = self.cutoff + self.skin
        periodic = [bool(b) for b in pbc] if (pbc is not None and cell is not None) else [False, False, False]
        # ---- coordinates the slabs and cells are cut in ----
        # With a cell: FRACTIONAL coordinates f_k = x . b_k (b_k: reciprocal vectors) for all three axes, periodic or not --
        # f_k does not change under a translation by the OTHER cell vectors, so the periodic images of an atom stay in its
        # layer (a Cartesian slab axis would not do when a periodic cell vector leans along it), and two atoms are at least
        # |delta f_k| x (spacing of the lattice planes) apart.  Without one: Cartesian axes.
        depth = [1.0, 1.0, 1.0]
        if cell is not None and any(periodic):
            c64 = cell.detach().to(torch.float64).cpu().reshape(3, 3)   # (host read of nine numbers)
            # x = f C  ->  f_k = x . (reciprocal vector k); closed form for 3 x 3 (no solver library)
            cr = torch.stack([torch.linalg.cross(c64[1], c64[2]), torch.linalg.cross(c64[2], c64[0]),
                              torch.linalg.cross(c64[0], c64[1])])
            det = float((c64[0] * cr[0]).sum())
            rec = cr / det
            # spacing of the lattice planes along each axis: volume / area of the face spanned by the other two vectors
            depth = (abs(det) / torch.linalg.norm(cr, dim=1)).tolist()
            f = x @ rec.T.to(device=dev, dtype=torch.float32)
        else:
            periodic = [False, False, False]
            f = x
        lo_f, ext_f = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
        if not all(periodic):   # (bounding box of the open axes: a host read of six numbers)
            mm = torch.aminmax(f, dim=0)
            mn, mx = mm.min.cpu().to(torch.float64).tolist(), mm.max.cpu().to(torch.float64).tolist()
            for k in range(3):
                if not periodic[k]:
                    lo_f[k], ext_f[k] = mn[k], max(mx[k] - mn[k], 1e-9)
        # the slab axis: the one along which the system is deepest
        length = [depth[k] * ext_f[k] for k in range(3)]
        axis = max(range(3), key=lambda k: length[k])
        self.axis, self.periodic = axis, periodic[axis]
        # Sort key: layers of a quarter cutoff along the slab axis, and inside a layer cells of about one cutoff along the other
        # two axes -- the cell-sorted order SURVEY 8(e) asks for.  Atoms that are close in space end up close in memory
        # (neighbor gathers and force pushes hit nearby rows), while the order stays monotone in the layer index, so a
        # rank's halo is a contiguous run of layers before and behind its range.
        nb = [1, 1, 1]
        for k in range(3):
            width = 0.25 * self.cutoff if k == axis else self.cutoff
            nb[k] = int(max(1, min(1 << 10 if k != axis else 1 << 20, length[k] // max(width, 1e-6))))
        while nb[0] * nb[1] * nb[2] >= 1 << 31:   # (32-bit keys; a system a hundred kilometres long gets coarser cells)
            k = max((k for k in range(3) if k != axis), key=lambda k: nb[k])
            k = k if nb[k] > 1 else axis
            nb[k] = (nb[k] + 1) // 2
        # unit coordinates: periodic axes wrapped into [0, 1), open axes scaled by their bounding box
        per = torch.tensor(periodic, device=dev)
        u = (f - torch.tensor(lo_f, device=dev, dtype=torch.float32)) / torch.tensor(ext_f, device=dev, dtype=torch.float32)
        u = torch.where(per, u - torch.floor(u), u).clamp_(0.0, 1.0 - 1e-6)
        nbt = torch.tensor(nb, device=dev, dtype=torch.float32)
        kk = (u * nbt).to(torch.int32)
        kk = torch.minimum(kk, torch.tensor(nb, device=dev, dtype=torch.int32) - 1)
        o1, o2 = [k for k in range(3) if k != axis]
        kx = kk[:, axis]
        key = (kx * nb[o1] + kk[:, o1]) * nb[o2] + kk[:, o2]          # (< 2^20 * 2^10 * 2^10 / ... fits int32 for real boxes)
        if species is not None:                                      # padding atoms last: nobody's neighbors
            pad = species.reshape(-1) < 0
            key = torch.where(pad, torch.full_like(key, 0x7FFFFFFF), key)
            kx = torch.where(pad, torch.full_like(kx, 2 * nb[axis] + 2), kx)
        key, order = torch.sort(key, stable=True)
        kxs = kx[order]                                              # layer of every sorted position (non-decreasing)
        self.order = order                                           # sorted position -> input atom
        self.bounds = shard_bounds(n, world)
        # ---- halos, in whole layers: everything within ceil(reach / layer width) + 1 layers of the first / last owned layer ----
        nl_ = nb[axis]
        # (a neighbor of an atom in layer k lies in layer >= k - ceil(reach / width): whole layers, plus a hair for the fp32 binning)
        dl = int(min(nl_, math.ceil(reach * nl_ / max(length[axis], 1e-9) + 1e-3)))
        b = torch.tensor(self.bounds, device=dev)
        lo_i, hi_i = b[:-1], b[1:]
        n_real = torch.searchsorted(kxs, torch.tensor([nl_], device=dev, dtype=torch.int32), right=False)   # padding starts here
        # (layer bounds of a rank at its REAL positions: a range that ends in padding would otherwise take the padding
        # layer for its last layer and, under PBC, every remaining atom for its right halo)
        k_lo = kxs[torch.minimum(lo_i, n_real).clamp(max=n - 1)].to(torch.int64)
        k_hi = kxs[(torch.minimum(hi_i, n_real) - 1).clamp(min=0)].to(torch.int64)
        q = torch.cat([k_lo - dl, k_lo - dl + nl_, k_hi + dl, k_hi + dl - nl_]).clamp(min=-1, max=2 * nl_ + 1).to(torch.int32)
        left = torch.searchsorted(kxs, q[:2 * world].contiguous(), right=False)
        right = torch.searchsorted(kxs, q[2 * world:].contiguous(), right=True)
        host = torch.cat([left, right, k_lo, k_hi, n_real]).cpu().tolist()   # (the second host sync of a partition)
        self.n_real = nr_ = int(host[-1])
        self.halo = []                                               # per rank: (n_left, n_right)
        for r in range(world):
            lo, hi = self.bounds[r], self.bounds[r + 1]
            lo_r, hi_r = min(lo, nr_), min(hi, nr_)                  # the rank's real (non-padding) positions
            own = hi_r - lo_r
            if own == 0:
                self.halo.append((0, 0))
                continue
            kl, kh = host[4 * world + r], host[5 * world + r]
            nl = lo_r - min(host[r], lo_r)
            nr = min(max(host[2 * world + r], hi_r), nr_) - hi_r
            if self.periodic:
                if kl - dl < 0:                                      # wraps below layer 0: everything from the wrapped layer up
                    nl = lo_r + (nr_ - min(host[world + r], nr_))
                if kh + dl >= nl_:
                    nr = (nr_ - hi_r) + min(host[3 * world + r], nr_)
            nl = min(nl, nr_ - own)
            nr = min(nr, nr_ - own - nl)
            self.halo.append((nl, nr))
        self.n_left, self.n_right = self.halo[rank]
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        self.n_owned = self.hi - self.lo
        self.n_local = self.n_left + self.n_owned + self.n_right
        m = max(nr_, 1)
        ar = torch.arange(max(self.n_left, self.n_right, 1), device=dev)
        pos = torch.cat([(ar[:self.n_left] + (min(self.lo, nr_) - self.n_left)) % m,
                         torch.arange(self.lo, self.hi, device=dev),
                         (ar[:self.n_right] + min(self.hi, nr_)) % m])
        self.local_pos = pos                                         # local row -> sorted position
        self.local_idx = order[pos]                                  # local row -> input atom
        self.owned_idx = self.local_idx[self.n_left:self.n_left + self.n_owned]
        self.x_build = x.clone() if self.skin > 0.0 else None
        self._cell_build = None if cell is None else cell.detach().clone()
        # ---- who holds halo rows of whom: runs (holder, first halo row of the holder, owner, first owned row, count) ----
        self.halo_rows_max = max((a + c for a, c in self.halo), default=0)
        self.messages = self._plan()
        # Exchange tables: a rank SENDS the halo rows it holds to the rank that owns them and to nobody else (for slabs
        # these are its one or two neighbours in the slab order), message after message in the order of the plan; it
        # RECEIVES, from every holder of its atoms, rows in the order that holder sends them.
        send, dst = [[] for _ in range(world)], [[] for _ in range(world)]
        for holder, hrow, owner, orow, cnt in self.messages:
            if cnt <= 0:
                continue
            if holder == rank:   # halo row h is local row h (left part) or n_owned + h (right part)
                h = torch.arange(hrow, hrow + cnt, device=dev)
                send[owner].append(torch.where(h < self.n_left, h, h + self.n_owned))
            if owner == rank:
                dst[holder].append(torch.arange(cnt, device=dev) + (self.n_left + orow))
        empty = torch.zeros(0, dtype=torch.long, device=dev)
        self.send_counts = [int(sum(t.numel() for t in send[o])) for o in range(world)]    # rows for owner o
        self.recv_counts = [int(sum(t.numel() for t in dst[h])) for h in range(world)]     # rows from holder h
        self.send_rows = torch.cat([t for o in range(world) for t in send[o]] or [empty])  # local rows, in send order
        self.recv_dst = torch.cat([t for h in range(world) for t in dst[h]] or [empty])    # local rows, in receive order
        self.peers = sorted({o for o in range(world) if o != rank and (self.send_counts[o] or self.recv_counts[o])})
        # (where the received row words and the tails sit in the receive buffer [rows of holder 0 | tail 0 | rows of holder 1 | ...],
        # as device tensors, so that unpacking needs no host work that depends on device data)
        rc = torch.tensor(self.recv_counts, dtype=torch.long)
        self._recv_word_holder = torch.repeat_interleave(torch.arange(world), 3 * rc).to(dev)
        self._recv_rows_end = (3 * rc).cumsum(0).to(dev)
        self._check = None        # pending asynchronous validity check (check_async)

    def still_valid(self, coords: torch.Tensor) -> bool:
        """Has every atom stayed within skin / 2 of where it was when the partition was built?  (One reduction + one host
        sync; always False for partitions built without a skin.)"""
        if self.x_build is None:
            return False
        d = coords.detach().reshape(-1, 3).to(torch.float32) - self.x_build
        return bool(((d * d).sum(dim=1).max() < (0.5 * self.skin) ** 2).item())

    # The per-step question "is the partition still good?" without a host synchronisation in the steady state: the flags
    # are computed on the device, copied to pinned host memory behind an event, and READ ONE STEP LATE (like the neighbor
    # builder's overflow word).  Two thresholds make the lag safe: the partition is renewed as soon as an atom has moved
    # SOFT x skin / 2 (0.8: rebuilt at the next step), and a step is only WRONG once an atom has moved skin / 2 itself -- the
    # step in between is covered as long as no atom moves 0.1 skin in one step (0.1 A for the default 1 A skin: twenty
    # times a hydrogen's motion in a 0.5 fs step); if it did, the late read raises instead of returning silently.
    SOFT = 0.8

    def _flags(self, coords: torch.Tensor, cell: tp.Optional[torch.Tensor]) -> torch.Tensor:
        """int32[2] on the device: word 0 = some atom moved SOFT x skin / 2 since the cut (or the cell changed), word 1 = some
        atom moved skin / 2."""
        d = coords.detach().reshape(-1, 3).to(torch.float32) - self.x_build
        # effective displacement: the largest motion of an atom, plus half of what the periodic images moved with the cell
        # (two atoms each d apart from where they were and an image shifted by |delta cell| change a distance by <= 2 d + |delta cell|)
        d_eff = (d * d).sum(dim=1).max().sqrt()
        changed = torch.zeros((), dtype=torch.bool, device=d_eff.device)
        if cell is not None and self._cell_build is not None:
            dc = cell.detach().to(self._cell_build.dtype).reshape(3, 3) - self._cell_build.reshape(3, 3)
            d_eff = d_eff + 0.5 * torch.linalg.norm(dc, dim=1).sum().to(d_eff.dtype)
            changed = (dc != 0).any()
        lim = 0.5 * self.skin
        return torch.stack([(d_eff >= lim * self.SOFT) | changed, d_eff >= lim]).to(torch.int32)

    def check_now(self, coords: torch.Tensor, cell: tp.Optional[torch.Tensor]) -> tp.Tuple[bool, bool]:
        """(renew, invalid) for ``coords`` themselves, read at once (one host synchronisation): the same-step guard of callers
        that are not in a steady loop -- the first moved step after a cut, ``ANI.partition_check = "strict"``."""
        if self.x_build is None:
            return True, True
        f = self._flags(coords, cell).tolist()
        return bool(f[0]), bool(f[1])

    def check_async(self, coords: torch.Tensor, cell: tp.Optional[torch.Tensor]) -> None:
        """Queue the validity flags for ``coords`` on the current stream: word 0 = some atom moved SOFT x skin / 2 since the
        cut (or the cell changed) -> renew the partition at the next step; word 1 = some atom moved skin / 2 -> the step that
        used this partition with these coordinates cannot be trusted.  No host synchronisation: ``poll`` reads them later."""
        if self.x_build is None:
            self._check = None
            return
        flags = self._flags(coords, cell)
        if flags.is_cuda:
            host = torch.empty(2, dtype=torch.int32, pin_memory=True)
            host.copy_(flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._check = (host, ev)
        else:
            self._check = (flags, None)

    @property
    def check_pending(self) -> bool:
        return self._check is not None

    def poll(self) -> tp.Tuple[bool, bool]:
        """(renew, invalid) of the LAST ``check_async`` (False, False if none is pending).  Waits for that check's event only
        -- work queued a step ago, long finished unless the host runs a whole step ahead of the device."""
        if self._check is None:
            return False, False
        host, ev = self._check
        self._check = None
        if ev is not None:
            ev.synchronize()
        return bool(host[0]), bool(host[1])

    def _plan(self) -> tp.List[tp.Tuple[int, int, int, int, int]]:
        """Halo rows of every rank cut into runs by the rank that owns them.  Halo row h of a holder (0 .. n_left + n_right,
        left halo first) is sorted position (lo - n_left + h) for the left part and (hi + h - n_left) for the right part,
        modulo the number of real atoms (padding atoms are sorted last and are in nobody's halo)."""
        m, out = max(self.n_real, 1), []
        for holder in range(self.world):
            nl, nr = self.halo[holder]
            lo, hi = min(self.bounds[holder], self.n_real), min(self.bounds[holder + 1], self.n_real)
            for first_pos, first_row, count in ((lo - nl, 0, nl), (hi, nl, nr)):
                done = 0
                while done < count:
                    p = (first_pos + done) % m
                    owner = max(0, min(self.world - 1, _bisect(self.bounds, p)))
                    run = min(count - done, min(self.bounds[owner + 1], self.n_real) - p)
                    out.append((holder, first_row + done, owner, p - self.bounds[owner], run))
                    done += run
        return out

    # ---- per-step data movement ----
    def local(self, t: torch.Tensor, width: int = 1) -> torch.Tensor:
        """Rows of a per-atom tensor ([N] or [N, width]) for the local system, in local order."""
        return t.reshape(self.n, -1)[self.local_idx].reshape((self.n_local,) if width == 1 else (self.n_local, width))

    def exchange(self, rows: torch.Tensor, tail: tp.Optional[torch.Tensor], group) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]:
        """rows [n_local, 3] (fp32 or int64): partial per-atom sums of this rank; tail: a small fp64 (or int64) vector to be
        SUMMED over the ranks (partial energies, virial).  ONE collective, an all-to-all with uneven pieces: a rank sends the
        halo rows it holds to the rank that owns them -- its neighbours in the slab order, nobody else gets force rows --
        and its tail to everybody (a few words; fp64 carried as pairs of fp32 words), so that every rank adds the same W
        partial sums in the same order.  Returns (rows with the other ranks' pushes onto this rank's owned atoms added,
        summed tail).  Bytes sent per rank: ``last_bytes``; ranks force rows travel to / from: ``peers``."""
        world = self.world
        nt = 0 if tail is None else tail.numel()
        wide = rows.dtype == torch.int64
        ntw = nt if wide else 2 * nt                       # words of the tail in the rows' dtype
        tw = None
        if nt:
            tw = tail.reshape(-1) if wide else tail.to(torch.float64).contiguous().view(torch.float32).reshape(-1)
        halo = rows[self.send_rows].reshape(-1)           # [3 * rows sent], grouped by owner
        pieces, off = [], 0
        for o in range(world):
            c = 3 * self.send_counts[o]
            pieces.append(halo[off:off + c])
            if nt:
                pieces.append(tw)
            off += c
        send = torch.cat(pieces) if pieces else rows.new_zeros(0)
        in_split = [3 * self.send_counts[o] + ntw for o in range(world)]
        out_split = [3 * self.recv_counts[h] + ntw for h in range(world)]
        got = _all_to_all(send, in_split, out_split, group)
        # unpack: [rows from holder 0 | tail 0 | rows from holder 1 | tail 1 | ...]
        total = None
        if nt:
            hh = torch.arange(world, device=got.device)
            idx = ((self._recv_rows_end + hh * ntw).view(-1, 1) + torch.arange(ntw, device=got.device).view(1, -1)).reshape(-1)
            parts = got[idx].view(world, ntw)
            parts = parts if wide else parts.contiguous().view(torch.float64).view(world, nt)
            total = parts.sum(dim=0)   # (the same W numbers in the same order on every rank: identical results)
        if self.recv_dst.numel():
            k = torch.arange(self._recv_word_holder.numel(), device=got.device)
            rows.index_add_(0, self.recv_dst, got[k + self._recv_word_holder * ntw].view(-1, 3))
        self.last_bytes = int(sum(in_split) - in_split[self.rank]) * send.element_size()
        return rows, total

    def scatter_owned(self, local_rows: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
        """[n_local, ...] -> [N, ...] in INPUT order holding this rank's owned rows (others ``fill``)."""
        out = torch.full((self.n,) + tuple(local_rows.shape[1:]), fill, dtype=local_rows.dtype, device=local_rows.device)
        out[self.owned_idx] = local_rows[self.n_left:self.n_left + self.n_owned]
        return out

    def scatter_local(self, local_rows: torch.Tensor) -> torch.Tensor:
        """[n_local, ...] -> [N, ...] in input order with EVERY local row (halo rows too): a rank's partial result before
        the exchange, as ``energies_and_forces(shard=(rank, world))`` returns it -- the partials of all ranks add up to
        the whole."""
        out = torch.zeros((self.n,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
        out[self.local_idx] = local_rows
        return out

    def gather_owned(self, local_rows: torch.Tensor, group) -> torch.Tensor:
        """All ranks' owned rows, [N, ...] in input order on every rank (one all-gather of equal, padded pieces)."""
        width = int(local_rows[0].numel()) if local_rows.dim() > 1 else 1
        m = max(self.bounds[r + 1] - self.bounds[r] for r in range(self.world))
        send = torch.zeros((m, width), dtype=local_rows.dtype, device=local_rows.device)
        send[:self.n_owned] = local_rows[self.n_left:self.n_left + self.n_owned].reshape(self.n_owned, width)
        got = _all_gather(send, self.world, group).view(self.world * m, width)
        srt = torch.cat([got[r * m:r * m + (self.bounds[r + 1] - self.bounds[r])] for r in range(self.world)])
        out = torch.empty_like(srt)
        out[self.order] = srt
        return out.reshape((self.n,) + tuple(local_rows.shape[1:]))


def _bisect(bounds: tp.Sequence[int], p: int) -> int:
    """Rank owning sorted position p (bounds monotone, empty ranks skipped)."""
    import bisect

    return bisect.bisect_right(bounds, p) - 1


def _all_gather(send: torch.Tensor, world: int, group) -> torch.Tensor:
    """all_gather_into_tensor of equal pieces, [world * len(send)].  RCCL ("nccl") gathers device tensors in place; the gloo
    backend (CPU tests, and the single-GPU development runs that put several ranks on one device) is staged through the
    host, where it implements the collective."""
    flat = send.reshape(-1)
    if send.is_cuda and torch.distributed.get_backend(group) == "gloo":
        host = torch.empty(world * flat.numel(), dtype=flat.dtype)
        torch.distributed.all_gather_into_tensor(host, flat.cpu(), group=group)
        return host.to(send.device)
    got = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    torch.distributed.all_gather_into_tensor(got, flat, group=group)
    return got


# How the uneven all-to-all of SpatialShards.exchange travels: "collective" = ONE all_to_all_single (RCCL: grouped
# point-to-point sends, only the non-empty pairs move data), "p2p" = the same pieces as paired isend / irecv in one batch
# (torch.distributed.batch_isend_irecv: an ncclGroup of sends and receives).  The transport is chosen ONCE per process, by a
# probe ahead of the first real exchange (_probe_transport): a tiny uneven all_to_all_single, then an all-reduce (MIN) of
# "it worked" so that every rank takes the same decision -- a rank-local failure can no longer leave one rank in isend / irecv
# while its peers wait in the collective (round-5 advice).  After the probe an error of the exchange is raised, not swallowed.
# TORCHANI_AMD_EXCHANGE=p2p forces the point-to-point form from the start (no probe).
_EXCHANGE = {"transport": os.environ.get("TORCHANI_AMD_EXCHANGE", "collective"), "fell_back": None,
             "probed": os.environ.get("TORCHANI_AMD_EXCHANGE") is not None}

# stand-in "group" of development runs on one GPU (bench.py --emulate-shard R/W --emulate-collective-bytes): the exchange packs
# and unpacks with the real byte plan of rank R and skips the wire (the received words are zeros)
EMULATE_WIRE = "emulate-wire"


def _probe_transport(group, device: torch.device, staged: bool) -> None:
    """Decide "collective" vs "p2p" for this process group, the same on every rank."""
    world = torch.distributed.get_world_size(group)
    rank = torch.distributed.get_rank(group)
    dev = torch.device("cpu") if staged else device
    ok, msg = 1, None
    try:
        # uneven on purpose: rank r sends 1 + (r + peer) % 2 words to every peer
        in_split = [1 + (rank + p) % 2 for p in range(world)]
        out_split = [1 + (p + rank) % 2 for p in range(world)]
        src = torch.zeros(sum(in_split), dtype=torch.float32, device=dev)
        got = torch.empty(sum(out_split), dtype=torch.float32, device=dev)
        torch.distributed.all_to_all_single(got, src, out_split, in_split, group=group)
    except (RuntimeError, ValueError, NotImplementedError) as err:   # (argument / support errors of the call)
        ok, msg = 0, f"{type(err).__name__}: {str(err)[:200]}"
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        _EXCHANGE["transport"] = "p2p"
        _EXCHANGE["fell_back"] = msg or "the probe all_to_all_single raised on another rank"
    _EXCHANGE["probed"] = True


def exchange_transport() -> tp.Dict[str, tp.Optional[str]]:
    """{"transport": "collective" | "p2p", "fell_back": the collective's error message if it raised}."""
    return dict(_EXCHANGE)


def _p2p_exchange(got: torch.Tensor, send: torch.Tensor, in_split: tp.Sequence[int], out_split: tp.Sequence[int], group) -> None:
    rank = torch.distributed.get_rank(group)
    world = torch.distributed.get_world_size(group)
    so = [0]
    for c in in_split:
        so.append(so[-1] + int(c))
    ro = [0]
    for c in out_split:
        ro.append(ro[-1] + int(c))
    got[ro[rank]:ro[rank + 1]] = send[so[rank]:so[rank + 1]]   # (own piece)
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        g_peer = torch.distributed.get_global_rank(group, peer) if group is not None else peer
        if out_split[peer]:
            ops.append(torch.distributed.P2POp(torch.distributed.irecv, got[ro[peer]:ro[peer + 1]], g_peer, group))
        if in_split[peer]:
            ops.append(torch.distributed.P2POp(torch.distributed.isend, send[so[peer]:so[peer + 1]], g_peer, group))
    if ops:
        for req in torch.distributed.batch_isend_irecv(ops):
            req.wait()


def _all_to_all(send: torch.Tensor, in_split: tp.Sequence[int], out_split: tp.Sequence[int], group) -> torch.Tensor:
    """all_to_all_single with uneven pieces (RCCL: grouped point-to-point sends, only the non-empty pairs move data), or the
    same pieces as batched isend / irecv (_EXCHANGE, chosen by _probe_transport ahead of the first exchange).  The gloo backend
    (CPU tests, single-GPU development runs with several ranks on one device) is staged through the host."""
    n_out = int(sum(out_split))
    if group is EMULATE_WIRE:   # (development: pack / unpack with the real byte plan, no wire)
        return torch.zeros(n_out, dtype=send.dtype, device=send.device)
    stage = send.is_cuda and torch.distributed.get_backend(group) == "gloo"
    if not _EXCHANGE["probed"]:
        _probe_transport(group, send.device, stage)
    src = send.cpu() if stage else send.contiguous()
    got = torch.empty(n_out, dtype=send.dtype, device=src.device)
    if _EXCHANGE["transport"] == "collective":
        torch.distributed.all_to_all_single(got, src, list(out_split), list(in_split), group=group)
    else:
        _p2p_exchange(got, src, in_split, out_split, group)
    return got.to(send.device) if stage else got
