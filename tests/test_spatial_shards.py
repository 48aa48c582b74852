"""Spatial shards (torchani_amd/parallel.py SpatialShards): host-side plan on CPU tensors, no engine needed.
  * the owned ranges partition the atoms for ANY input order, each is a slab along the chosen axis;
  * the local system of a rank contains every atom within the cutoff of its owned atoms (minimum image);
  * the message plan is consistent: what a holder keeps for an owner lands on the right owned rows;
  * a 2- and 3-rank gloo run of ``exchange`` reproduces the single-rank sums of random pair pushes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _box(n, L, seed, shuffle=True):
    rs = np.random.RandomState(seed)
    x = rs.uniform(0.0, 1.0, (n, 3)) * np.array(L)
    if not shuffle:
        x = x[np.argsort(x[:, int(np.argmax(L))])]
    return torch.from_numpy(x.astype(np.float32))


def _min_image_pairs(x, L, pbc, rc):
    d = x[:, None, :].double() - x[None, :, :].double()
    for k in range(3):
        if pbc[k]:
            d[..., k] -= L[k] * torch.round(d[..., k] / L[k])
    r2 = (d * d).sum(-1)
    m = r2 < rc * rc
    m.fill_diagonal_(False)
    return m


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("pbc", [(True, True, True), (False, False, False), (False, True, True)])
def test_plan_covers_neighbors_and_is_consistent(world, pbc):
    from torchani_amd.parallel import SpatialShards

    L = [40.0, 17.0, 15.0]
    n = 1500
    x = _box(n, L, 0)
    cell = torch.diag(torch.tensor(L))
    rc = 5.1
    near = _min_image_pairs(x, L, pbc, rc)
    parts = [SpatialShards(x, cell, pbc, world, r, rc) for r in range(world)]
    assert parts[0].axis == 0 and parts[0].periodic == pbc[0]   # the longest edge, with or without PBC along it
    owned_all = torch.cat([p.owned_idx for p in parts])
    assert sorted(owned_all.tolist()) == list(range(n))            # a partition of the atoms, whatever their order
    for p in parts:
        loc = set(p.local_idx.tolist())
        assert len(loc) == p.n_local                                # no atom twice in a local system
        need = torch.nonzero(near[p.owned_idx].any(dim=0)).reshape(-1).tolist()
        assert set(need) <= loc, "a neighbor of an owned atom is missing from the local system"
        assert p.messages == parts[0].messages and p.halo == parts[0].halo
    # every halo row of every holder appears in exactly one run, and the run points at the same atom on the owner's side
    for holder, hrow, owner, orow, cnt in parts[0].messages:
        h, o = parts[holder], parts[owner]
        lrow = hrow if hrow < h.n_left else h.n_owned + hrow        # halo row -> local row of the holder
        assert torch.equal(h.local_idx[lrow:lrow + cnt], o.local_idx[o.n_left + orow:o.n_left + orow + cnt])
    for p in parts:
        rows = sum(c for hd, _, _, _, c in p.messages if hd == p.rank)
        assert rows == p.n_left + p.n_right
    if world == 8 and all(pbc):
        assert max(p.n_local for p in parts) < 0.62 * n             # slabs: 1/8 owned + two (5.1 A + a layer) halos of a 40 A box


@pytest.mark.parametrize("periodic", [True, False])
def test_force_rows_travel_between_slab_neighbours_only(periodic):
    """Slabs thicker than the reach (here 15 A against 5.1 A + skin): a rank exchanges force rows with the rank before and
    the rank behind it in the slab order and with nobody else (SpatialShards.peers), and the async validity flags say
    "renew" at 0.8 x skin / 2 of motion and "invalid" at skin / 2 -- without the caller synchronising."""
    from torchani_amd.parallel import SpatialShards

    L = [120.0, 12.0, 12.0]
    x = _box(2400, L, 1)
    cell = torch.diag(torch.tensor(L))
    pbc = (periodic,) * 3
    world = 8
    parts = [SpatialShards(x, cell, pbc, world, r, 5.1, skin=1.0) for r in range(world)]
    for p in parts:
        want = {(p.rank - 1) % world, (p.rank + 1) % world} if periodic else {r for r in (p.rank - 1, p.rank + 1) if 0 <= r < world}
        assert set(p.peers) == want, (p.rank, p.peers)
        assert sum(c > 0 for c in p.send_counts) <= 2 and sum(c > 0 for c in p.recv_counts) <= 2
    p = parts[3]
    assert p.poll() == (False, False)                       # nothing pending
    p.check_async(x, cell)
    assert p.poll() == (False, False)                       # nobody moved
    y = x.clone()
    y[7, 1] += 0.45                                          # 0.45 A > 0.8 x 0.5 A: renew next step, this step still fine
    p.check_async(y, cell)
    assert p.poll() == (True, False)
    y[7, 1] += 0.1                                           # 0.55 A > skin / 2: the step that used the partition is not covered
    p.check_async(y, cell)
    assert p.poll() == (True, True)
    p.check_async(x, cell * 1.01)                            # a changed box always renews
    assert p.poll()[0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, transport="collective"):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        import torchani_amd.parallel as par
        from torchani_amd.parallel import SpatialShards

        if transport == "p2p":
            par._EXCHANGE["transport"], par._EXCHANGE["probed"] = "p2p", True
        elif transport == "fallback":   # the collective raises on its first call: every rank must switch to p2p and go on
            def broken(*a, **k):
                raise RuntimeError("all_to_all_single: uneven splits are not supported (simulated)")
            dist.all_to_all_single = broken

        dist.init_process_group("gloo")
        L = [30.0, 12.0, 12.0]
        n, rc = 600, 5.1
        x = _box(n, L, 3)                                           # shuffled input order
        cell = torch.diag(torch.tensor(L))
        pbc = (True, True, True)
        near = _min_image_pairs(x, L, pbc, rc)
        w = torch.from_numpy(np.random.RandomState(5).standard_normal((n, n, 3)).astype(np.float32))
        # "forces": central atom i pushes w[i, j] onto every neighbor j and -w[i, j] onto itself; energies e_i
        ref = torch.zeros(n, 3, dtype=torch.float64)
        for i in range(n):
            js = torch.nonzero(near[i]).reshape(-1)
            ref[js] += w[i, js].double()
            ref[i] -= w[i, js].double().sum(0)
        e_ref = float(torch.arange(n, dtype=torch.float64).mul(1e-3).add(0.123456789012345).sum())
        part = SpatialShards(x, cell, pbc, world, rank, rc)
        inv = {a: k for k, a in enumerate(part.local_idx.tolist())}
        rows = torch.zeros(part.n_local, 3, dtype=torch.float32)
        e_part = 0.0
        for a in part.owned_idx.tolist():
            js = torch.nonzero(near[a]).reshape(-1).tolist()
            for j in js:
                rows[inv[j]] += w[a, j]
                rows[inv[a]] -= w[a, j]
            e_part += a * 1e-3 + 0.123456789012345
        rows, tot = part.exchange(rows, torch.tensor([e_part], dtype=torch.float64), dist.group.WORLD)
        full = part.gather_owned(rows, dist.group.WORLD)
        mine = part.scatter_owned(rows)
        err = float((full.double() - ref).abs().max())
        err_own = float((mine[part.owned_idx].double() - ref[part.owned_idx]).abs().max())
        q.put((rank, {"err": err, "err_own": err_own, "dE": abs(float(tot[0]) - e_ref), "bytes": part.last_bytes,
                      "n_local": part.n_local, "transport": par.exchange_transport()}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback

        q.put((rank, {"error": repr(e) + "\n" + traceback.format_exc()}))


@pytest.mark.parametrize("world,transport", [(2, "collective"), (3, "collective"), (3, "p2p"), (2, "fallback")])
def test_exchange_over_gloo_equals_single_rank(world, transport):
    """... through ONE all_to_all_single, through the same pieces as batched isend / irecv, and when the collective raises on
    its first call (every rank falls back to isend / irecv and says so)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert "error" not in res[r], res[r]["error"]
        assert res[r]["err"] < 1e-4 and res[r]["err_own"] < 1e-4 and res[r]["dE"] < 1e-9, res[r]   # (fp32 sums of ~80 N(0, 1) pushes)
        tr = res[r]["transport"]
        assert tr["transport"] == ("collective" if transport == "collective" else "p2p")
        assert (tr["fell_back"] is not None) == (transport == "fallback")


def test_plan_triclinic_cell():
    """A sheared cell: neighbors by brute force over the 27 images; the slab axis is the one with the deepest lattice planes."""
    from torchani_amd.parallel import SpatialShards

    cell = torch.tensor([[34.0, 0.0, 0.0], [6.0, 14.0, 0.0], [-3.0, 4.0, 13.0]])
    n, rc, world = 900, 5.1, 4
    f = torch.from_numpy(np.random.RandomState(4).uniform(0, 1, (n, 3)).astype(np.float32))
    x = f @ cell
    d = x[:, None, :].double() - x[None, :, :].double()
    near = torch.zeros((n, n), dtype=torch.bool)
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            for c in (-1, 0, 1):
                sh = (a * cell[0] + b * cell[1] + c * cell[2]).double()
                near |= ((d + sh) ** 2).sum(-1) < rc * rc
    near.fill_diagonal_(False)
    parts = [SpatialShards(x, cell, (True, True, True), world, r, rc) for r in range(world)]
    assert parts[0].axis == 0
    assert sorted(torch.cat([p.owned_idx for p in parts]).tolist()) == list(range(n))
    for p in parts:
        need = torch.nonzero(near[p.owned_idx].any(dim=0)).reshape(-1).tolist()
        assert set(need) <= set(p.local_idx.tolist())
        assert p.n_local < n


def test_padding_atoms_are_in_nobodys_halo():
    from torchani_amd.parallel import SpatialShards

    """Padding atoms (species -1) sort last; a periodic wrap of the first rank's left halo must reach the last REAL atoms."""
    torch.manual_seed(3)
    n, box, rc = 600, 30.0, 5.1
    x = torch.rand(n, 3) * box
    sp = torch.randint(0, 4, (n,))
    sp[torch.randperm(n)[:50]] = -1
    cell = torch.eye(3) * box
    for world in (2, 3):
        parts = [SpatialShards(x, cell, (True, True, True), world, r, rc, species=sp) for r in range(world)]
        for part in parts:
            assert part.n_real == n - 50
            halo = torch.cat([part.local_idx[:part.n_left], part.local_idx[part.n_left + part.n_owned:]])
            assert bool((sp[halo] >= 0).all())
            # every real neighbor of an owned real atom is in the local system
            own = part.owned_idx[sp[part.owned_idx] >= 0]
            d = x[own][:, None, :] - x[None, :, :]
            d = d - box * torch.round(d / box)
            near = ((d * d).sum(-1) < rc * rc) & (sp[None, :] >= 0)
            have = torch.zeros(n, dtype=torch.bool)
            have[part.local_idx] = True
            assert bool(have[near.any(dim=0)].all())
        assert sorted(torch.cat([p.owned_idx for p in parts]).tolist()) == list(range(n))


def test_skin_keeps_the_partition_valid_while_atoms_move():
    from torchani_amd.parallel import SpatialShards

    torch.manual_seed(4)
    n, box, rc, skin = 800, 32.0, 5.1, 1.0
    x = torch.rand(n, 3) * box
    cell = torch.eye(3) * box
    part = SpatialShards(x, cell, (True, True, True), 3, 1, rc, skin=skin)
    assert not SpatialShards(x, cell, (True, True, True), 3, 1, rc).still_valid(x)
    step = torch.nn.functional.normalize(torch.randn(n, 3), dim=1) * (0.49 * skin)
    moved = x + step
    assert part.still_valid(moved)
    assert not part.still_valid(x + 1.1 * step)
    # with every atom moved by just under skin / 2, every neighbor of an owned atom is still in the local system
    own = part.owned_idx
    d = moved[own][:, None, :] - moved[None, :, :]
    d = d - box * torch.round(d / box)
    near = (d * d).sum(-1) < rc * rc
    have = torch.zeros(n, dtype=torch.bool)
    have[part.local_idx] = True
    assert bool(have[near.any(dim=0)].all())


@pytest.mark.parametrize("seed", [0, 1])
def test_random_boxes_cells_and_pbc(seed):
    """Randomised sweep over sizes (1 .. 600 atoms), world sizes, open / periodic / MIXED boundary conditions, triclinic cells
    (a periodic cell vector leaning along an open axis is what made Cartesian slabs wrong), planar systems, padding atoms
    and skins: every atom is owned once, every neighbor of an owned atom -- through any periodic image -- is in the local
    system, padding atoms are in nobody's halo."""
    from torchani_amd.parallel import SpatialShards

    rs = np.random.RandomState(seed)
    for trial in range(120):
        n = int(rs.choice([1, 2, 3, 7, 40, 200, 600]))
        world = int(rs.choice([1, 2, 3, 5, 8]))
        rc = float(rs.choice([2.0, 5.1]))
        skin = float(rs.choice([0.0, 0.0, 1.0]))
        if rs.rand() < 0.6:
            L = rs.uniform(2 * rc + 0.5, 30.0, 3)
            cell = np.diag(L)
            if rs.rand() < 0.5:
                cell[1, 0] = rs.uniform(-0.3, 0.3) * L[0]
                cell[2, 0] = rs.uniform(-0.3, 0.3) * L[0]
                cell[2, 1] = rs.uniform(-0.3, 0.3) * L[1]
            x = rs.uniform(-0.2, 1.2, (n, 3)) @ cell
            pbc = tuple(bool(b) for b in (rs.rand(3) < 0.7)) if rs.rand() < 0.5 else (True, True, True)
            pbc = pbc if any(pbc) else (True, True, True)
        else:
            cell, pbc = None, None
            ext = rs.uniform(0.0, 25.0, 3)
            if rs.rand() < 0.2:
                ext[rs.randint(3)] = 0.0
            x = rs.uniform(0, 1, (n, 3)) * ext
        sp = rs.randint(0, 4, n)
        if rs.rand() < 0.4 and n > 2:
            sp[rs.rand(n) < 0.2] = -1
        xt = torch.from_numpy(x.astype(np.float32))
        ct = None if cell is None else torch.from_numpy(cell.astype(np.float32))
        parts = [SpatialShards(xt, ct, pbc, world, r, rc, species=torch.from_numpy(sp), skin=skin) for r in range(world)]
        assert sorted(torch.cat([p.owned_idx for p in parts]).tolist()) == list(range(n))
        shifts = [np.zeros(3)]
        if cell is not None:
            rng = [(-1, 0, 1) if pbc[k] else (0,) for k in range(3)]
            shifts = [a * cell[0] + b * cell[1] + c * cell[2] for a in rng[0] for b in rng[1] for c in rng[2]]
        real = sp >= 0
        for p in parts:
            have = np.zeros(n, bool)
            have[p.local_idx.numpy()] = True
            own = p.owned_idx.numpy()
            own = own[real[own]]
            if own.size == 0:
                continue
            near = np.zeros(n, bool)
            for sh in shifts:
                d = x[None, :, :] + sh[None, None, :] - x[own][:, None, :]
                near |= ((d * d).sum(-1) <= rc * rc).any(axis=0)
            assert have[near & real].all(), (trial, n, world, pbc)
            halo = np.concatenate([p.local_idx[:p.n_left].numpy(), p.local_idx[p.n_left + p.n_owned:].numpy()])
            assert real[halo].all()
        # the exchange, emulated without a process group: what every rank pushed onto an atom arrives at its owner
        if world > 1:
            rows = [torch.from_numpy(rs.normal(size=(p.n_local, 3)).astype(np.float32)) for p in parts]
            want = torch.zeros(n, 3)
            sent = {}   # (holder, owner) -> the rows the holder sends to the owner, in its send order
            for p, rw in zip(parts, rows):
                want.index_add_(0, p.local_idx, rw)
                assert sum(p.send_counts) == p.n_left + p.n_right == p.send_rows.numel()
                off = 0
                for o in range(world):
                    sent[(p.rank, o)] = rw[p.send_rows[off:off + p.send_counts[o]]]
                    off += p.send_counts[o]
            for p, rw in zip(parts, rows):
                assert p.messages == parts[0].messages
                # every rank receives what the others send it, nothing else, and force rows only travel between ranks
                # whose slabs touch (the rank's neighbours in the slab order; everybody when the slabs are thinner than the reach)
                assert p.recv_counts == [parts[h].send_counts[p.rank] for h in range(world)]
                assert all(o != p.rank for o in p.peers)
                got = torch.cat([sent[(h, p.rank)] for h in range(world)])
                out = rw.clone()
                if p.recv_dst.numel():
                    out.index_add_(0, p.recv_dst, got)
                assert torch.allclose(out[p.n_left:p.n_left + p.n_owned], want[p.owned_idx], atol=1e-5)
