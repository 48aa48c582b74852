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
    _, gc = o.aev(p, g["species"], x, g["cell"], g["pbc"], grad_aev=ga)
    sp = g["species"].reshape(-1)
    e_part = float(sum(ae[i] + sae[sp[i]] for i in range(lo, hi)))
    energies = torch.tensor([e_part], dtype=torch.float64)
    forces = torch.from_numpy(-gc.astype(np.float32))
    dist.all_reduce(energies, group=group)
    dist.all_reduce(forces, group=group)
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
