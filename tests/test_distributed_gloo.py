"""world_size-2 test of the sharded energy+forces reduction on CPU with the gloo backend.

The HIP engine cannot run without a GPU, so the per-shard evaluation is played by the CPU oracle (the
test-infrastructure checker); what is under test is the product's decomposition logic in
torchani_amd.parallel + the collectives: contiguous central-atom shards, replicated coordinates, fp64
energy all-reduce, fp32 force all-reduce -- exactly what ANI.energies_and_forces does per rank.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _util import load_golden, oracle_networks, oracle_params
    from oracle.oracle import Oracle
    from torchani_amd.parallel import init_from_env, shard_range

    r, w, _, group = init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and group is not None
    g = load_golden("water_pbc_ani2x")
    C, A = g["species"].shape
    n = C * A
    lo, hi = shard_range(n, group)
    o = Oracle("f64")
    o.set_threads(1)
    p = oracle_params("ani2x")
    dims, flat, sae = oracle_networks("ani2x", 8, g["seed"])
    x = g["coords"].astype(np.float64)
    # this rank's shard: AEVs of all atoms (coords replicated), networks + backward only for [lo, hi)
    aev = o.aev(p, g["species"], x, g["cell"], g["pbc"])
    ae, ga, _ = o.mlp(g["species"], aev, dims, flat, n_members=8)
    mask = np.zeros(n, dtype=bool)
    mask[lo:hi] = True
    ga = np.where(mask[:, None], ga.reshape(n, -1), 0.0)
    _, gc, vir = o.aev(p, g["species"], x, g["cell"], g["pbc"], grad_aev=ga, want_virial=True)
    sp = g["species"].reshape(-1)
    e_part = float(sum(ae[i] + sae[sp[i]] for i in range(lo, hi)))
    energies = torch.tensor([e_part], dtype=torch.float64)
    forces = torch.from_numpy(-gc.astype(np.float32))
    virial = torch.from_numpy(vir)   # partial virial of this shard's central atoms: shards add up (fdotr)
    dist.all_reduce(energies, group=group)
    dist.all_reduce(forces, group=group)
    dist.all_reduce(virial, group=group)
    from _util import load_stress

    assert np.abs(virial.numpy() - load_stress("water_pbc_ani2x")["virial"]).max() < 1e-8
    err_e = abs(float(energies[0]) - float(g["energies"][0]))
    err_f = float(np.abs(forces.numpy().astype(np.float64) - g["forces"]).max())
    q.put((rank, lo, hi, err_e, err_f))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_energy_forces_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 30  # contiguous cover of the atoms
    for _, _, _, err_e, err_f in res:
        assert err_e < 1e-8 and err_f < 1e-6  # both ranks hold the reduced result


def _train_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _util import load_golden, load_wgrads, oracle_networks, oracle_params, wgrad_digest, wgrad_upstream
    from oracle.oracle import Oracle
    from torchani_amd.parallel import all_reduce_gradients, init_from_env, shard_bounds

    r, w, _, group = init_from_env(backend="gloo")
    g = load_golden("rand_batch_ani2x")
    C, A = g["species"].shape
    b = shard_bounds(C, w)              # the minibatch is split over molecules
    c0, c1 = b[r], b[r + 1]
    o = Oracle("f64")
    o.set_threads(2)
    p = oracle_params("ani2x")
    dims, flat, _ = oracle_networks("ani2x", 8, g["seed"])
    sp = g["species"][c0:c1]
    aev = o.aev(p, sp, g["coords"][c0:c1].astype(np.float64))
    up = wgrad_upstream(C, A)[c0:c1]    # the per-atom loss weights of the full batch, this rank's rows
    part = o.mlp_weight_grads(sp, aev, up, dims, flat, n_members=8)
    # parameters with this rank's gradients, in uneven chunks to exercise the bucket packing; one of them has no .grad
    cuts = np.linspace(0, part.shape[0], 40).astype(np.int64)
    params = []
    for i in range(len(cuts) - 1):
        t = torch.nn.Parameter(torch.zeros(int(cuts[i + 1] - cuts[i]), dtype=torch.float64))
        if not (i == 7 and np.all(part[cuts[i]:cuts[i + 1]] == 0)):
            t.grad = torch.from_numpy(part[cuts[i]:cuts[i + 1]].copy())
        params.append(t)
    all_reduce_gradients(params, group)
    total = np.concatenate([t.grad.numpy() for t in params])
    ref = load_wgrads("rand_batch_ani2x")
    sums, dots, heads = wgrad_digest(total)
    scale = float(ref["grad_abs_max"])
    err = max(np.abs(sums - ref["block_sums"]).max(), np.abs(dots - ref["block_dots"]).max(),
              np.abs(heads - ref["block_heads"]).max())
    # the same gradients as views of ONE flat buffer, tagged the way torchani_amd.optim.Adam tags its parameters (the
    # optimizer itself needs a ROCm device): the buffer is the bucket, reduced in place
    import weakref

    from torchani_amd.parallel import _one_flat_group

    class Flat:   # (what _one_flat_group reads of optim._FlatGroup)
        pass

    fg = Flat()
    fg.grad = torch.from_numpy(part.copy())
    fg.params, fg.grad_views = [], []
    for i in range(len(cuts) - 1):
        t = torch.nn.Parameter(torch.zeros(int(cuts[i + 1] - cuts[i]), dtype=torch.float64))
        v = fg.grad[int(cuts[i]):int(cuts[i + 1])]
        t.grad = v
        t._anihip_flat = (weakref.ref(fg), i)
        fg.params.append(t)
        fg.grad_views.append(v)
    assert _one_flat_group(fg.params) is fg and _one_flat_group(fg.params[:-1]) is None and _one_flat_group(params) is None
    ptr = fg.grad.data_ptr()
    all_reduce_gradients(fg.params, group)
    assert fg.grad.data_ptr() == ptr and all(t.grad is v for t, v in zip(fg.params, fg.grad_views))
    err_flat = float(np.abs(fg.grad.numpy() - total).max())
    q.put((rank, c0, c1, float(err / scale), err_flat / scale))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_gradients_gloo():
    """Data-parallel training step: molecules split over 2 ranks, per-rank weight gradients, ONE bucketed all-reduce
    (torchani_amd.parallel.all_reduce_gradients) -> the full-batch gradient of the reference's autograd
    (tests/golden/wgrads_rand_batch_ani2x.npz)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 6   # contiguous cover of the 6 molecules
    for _, _, _, rel, rel_flat in res:
        assert rel < 1e-7 and rel_flat < 1e-15   # (the flat route sums the same two numbers per element)
