"""The N > 1 PRODUCT path on one GPU: two ranks (gloo backend, both on cuda:0) call the real
``model.energies_and_forces(group=...)`` / ``all_reduce_gradients`` / ``bench.py --gpus 2`` and must reproduce the
single-rank results.  (The reference has no multi-device code; the decomposition is
torchani_amd/parallel.py.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F_TOL = 1e-4
E_ATOM_TOL = 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from torchani_amd.parallel import init_from_env

    r, w, _, group = init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and group is not None
    return group


def _worker_ef(rank, world, port, q):
    try:
        group = _setup(rank, world, port)
        import torch.distributed as dist
        from _util import load_golden, load_stress, seeded_state
        from bench import water_box
        from torchani_amd.models import ANI2x

        dev = torch.device("cuda", 0)
        out = {}
        # (a) golden periodic water fixture, batch-mode neighbor rows, with the virial
        g = load_golden("water_pbc_ani2x")
        model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False)
        sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
        x = torch.from_numpy(g["coords"]).to(dev)
        cell = torch.from_numpy(g["cell"]).to(dev)
        pbc = tuple(bool(b) for b in g["pbc"])
        single = model.energies_and_forces(sp, x, cell, pbc, stress=True)
        shard = model.energies_and_forces(sp, x, cell, pbc, group=group, stress=True)
        # (spatial shards: one all-gather of halo rows + partial energies, one more to hand every rank all forces)
        assert model.last_collective["collectives_per_step"] == 2 and model.last_collective["world_size"] == world
        out["golden_dE"] = float((shard.energies - single.energies).abs().max())
        out["golden_dF"] = float((shard.forces - single.forces).abs().max())
        out["golden_dW"] = float((shard.virial - single.virial).abs().max())
        out["golden_F_vs_ref"] = float(np.abs(shard.forces.cpu().numpy() - g["forces"]).max())
        out["golden_E_vs_ref"] = float(np.abs(shard.energies.cpu().numpy() - g["energies"]).max())
        out["golden_W_vs_ref"] = float(np.abs(shard.virial.cpu().numpy() - load_stress("water_pbc_ani2x")["virial"]).max())
        # (b) a 3000-atom water box through the cell list: atoms of a shard push onto atoms of the other one
        sp_np, x_np, cell_np = water_box(10)
        model2 = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
        sp2, x2, cell2 = torch.from_numpy(sp_np).to(dev), torch.from_numpy(x_np).to(dev), torch.from_numpy(cell_np).to(dev)
        s1 = model2.energies_and_forces(sp2, x2, cell2, (True, True, True), stress=True)
        s2 = model2.energies_and_forces(sp2, x2, cell2, (True, True, True), group=group, stress=True)
        out["box_dE"] = float((s2.energies - s1.energies).abs().max())
        out["box_dF"] = float((s2.forces - s1.forces).abs().max())
        out["box_dW"] = float((s2.virial - s1.virial).abs().max())
        out["box_bytes"] = model2.last_collective["bytes"]
        out["box_local"] = (model2.last_collective["n_local"], model2.last_collective["n_owned"])
        # the same box with its atoms in random order: identical shards of space, identical result
        perm = torch.from_numpy(np.random.RandomState(3).permutation(sp2.shape[1])).to(dev)
        s3 = model2.energies_and_forces(sp2[:, perm].contiguous(), x2[:, perm].contiguous(), cell2, (True, True, True),
                                        group=group, stress=True)
        out["shuffled_dE"] = float((s3.energies - s1.energies).abs().max())
        out["shuffled_dF"] = float((s3.forces - s1.forces[:, perm]).abs().max())
        # forces left with their owners: this rank's rows are complete, the others' are zero; one collective
        s4 = model2.energies_and_forces(sp2, x2, cell2, (True, True, True), group=group, reduce_forces=False)
        own = (s4.forces.abs().sum(dim=2) > 0).reshape(-1)
        out["owned_dF"] = float((s4.forces[0, own] - s1.forces[0, own]).abs().max())
        out["owned_n"] = int(own.sum())
        out["owned_coll"] = model2.last_collective["collectives_per_step"]
        # the round-2 scheme (index ranges + one all-reduce of the whole force array) stays selectable
        model2.partition = "index"
        s5 = model2.energies_and_forces(sp2, x2, cell2, (True, True, True), group=group, stress=True)
        out["index_dF"] = float((s5.forces - s1.forces).abs().max())
        out["index_bytes"] = model2.last_collective["bytes"]
        model2.partition = "spatial"
        # deterministic mode: ONE int64 all-reduce (forces, energy and virial in 2^-32 fixed point); a sharded run is
        # bit-reproducible (integer sums do not care who adds what, or in which order the ranks are reduced) and equals
        # the single-rank result to the rounding of the pairs that straddle the shards
        model2.deterministic_forces = True
        d1 = model2.energies_and_forces(sp2, x2, cell2, (True, True, True), stress=True)
        d2 = model2.energies_and_forces(sp2, x2, cell2, (True, True, True), group=group, stress=True)
        f2 = d2.forces.clone()
        d3 = model2.energies_and_forces(sp2, x2, cell2, (True, True, True), group=group, stress=True)
        out["det_equal"] = bool(torch.equal(f2, d3.forces))
        out["det_vs_single"] = float((d1.forces - f2).abs().max())
        out["det_dE"] = float((d1.energies - d2.energies).abs().max())
        out["det_bytes"] = model2.last_collective["bytes"]
        model2.deterministic_forces = False
        # every rank holds the same reduced result
        chk = torch.stack([s2.energies.sum(), s2.forces.double().abs().sum(), s2.virial.sum()]).cpu()
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk, group=group)
        out["rank_spread"] = float((both[0] - both[1]).abs().max())
        # (d) potentials wider than the AEV's cutoff on spatial shards: ANI-2dr = networks + xTB repulsion (5.2 A) + DFT-D3
        # dispersion (8 A, coordination numbers: a halo of three cutoffs), on a box three times as long as it is wide so
        # that the 24 A halos do not simply cover everything.  ONE collective; D3 is evaluated for the local system only.
        from torchani_amd.models import ANI2dr

        spw, xw, cw = water_box(12)
        L0 = float(cw[0, 0])
        xs = np.concatenate([xw + np.array([k * L0, 0.0, 0.0], dtype=np.float32) for k in range(3)], axis=1)
        sps = np.concatenate([spw] * 3, axis=1)
        cells = cw.copy()
        cells[0, 0] = 3 * L0
        m2d = ANI2dr(seed=5, device=dev, periodic_table_index=False, neighborlist="cell", row_capacity=160)
        spd, xd, celld = torch.from_numpy(sps).to(dev), torch.from_numpy(xs).to(dev), torch.from_numpy(cells).to(dev)
        d_one = m2d.energies_and_forces(spd, xd, celld, (True, True, True), stress=True)
        d_two = m2d.energies_and_forces(spd, xd, celld, (True, True, True), group=group, reduce_forces=False, stress=True)
        own = (d_two.forces.abs().sum(dim=2) > 0).reshape(-1)
        out["d3_coll"] = m2d.last_collective["collectives_per_step"]
        out["d3_op"] = m2d.last_collective["op"]
        out["d3_local"] = (m2d.last_collective["n_local"], m2d.last_collective["n_owned"], int(spd.numel()))
        out["d3_dE"] = float((d_two.energies - d_one.energies).abs().max())
        out["d3_dF"] = float((d_two.forces[0, own] - d_one.forces[0, own]).abs().max())
        out["d3_dW"] = float((d_two.virial - d_one.virial).abs().max())
        out["d3_Fmax"] = float(d_one.forces.abs().max())
        # (c) batch of molecules that do not straddle ranks: only the 16-byte-per-molecule energy tail is reduced
        gb = load_golden("rand_batch_ani2x")
        modelb = ANI2x(state_dict=seeded_state("ani2x", 8, gb["seed"]), device=dev, periodic_table_index=False)
        spb = torch.from_numpy(gb["species"].astype(np.int64)).to(dev)
        xb = torch.from_numpy(gb["coords"]).to(dev)
        assert spb.shape[0] % world == 0
        b1 = modelb.energies_and_forces(spb, xb)
        b2 = modelb.energies_and_forces(spb, xb, group=group, reduce_forces=False)
        lo, hi = rank * spb.shape[0] // world, (rank + 1) * spb.shape[0] // world
        out["batch_dE"] = float((b2.energies - b1.energies).abs().max())
        out["batch_dF_own"] = float((b2.forces[lo:hi] - b1.forces[lo:hi]).abs().max())
        out["batch_bytes"] = modelb.last_collective["bytes"]
        out["batch_C"] = int(spb.shape[0])
        torch.cuda.synchronize()
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:   # surface the failure in the parent instead of a queue timeout
        import traceback

        q.put((rank, {"error": repr(e) + "\n" + traceback.format_exc()}))


def _run(worker, world=2, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert "error" not in res[r], res[r]["error"]
    return res


def test_two_ranks_energies_forces_virial_match_single_rank():
    res = _run(_worker_ef)
    for r, o in res.items():
        # sharded == unsharded to fp32 round-off (float atomics reorder the sums), energies exactly summed
        assert o["golden_dE"] < 5e-8 and o["box_dE"] < 2e-7, o   # (other tiles of atoms share a power-of-two scale: ~1e-9 per atom)
        assert o["golden_dF"] < 2e-6 and o["box_dF"] < 2e-6, o
        assert o["golden_dW"] < 1e-6 and o["box_dW"] < 1e-5, o
        # and equal to the reference fixture
        assert o["golden_F_vs_ref"] < F_TOL and o["golden_E_vs_ref"] < E_ATOM_TOL * 10 and o["golden_W_vs_ref"] < 1e-5, o
        assert o["rank_spread"] == 0.0, o
        assert o["det_equal"], o
        assert o["det_vs_single"] < 2e-7 and o["det_dE"] < 5e-7, o
        assert o["box_local"][1] == 1500 and o["box_local"][0] <= 3000, o
        # (total energy of 3000 atoms in other tiles, hence other power-of-two scales and another order of the partial sums:
        # 1.8e-7 Ha with the 16 x 16 x 32 tiles of round 6 = 6e-11 Ha per atom)
        assert o["shuffled_dE"] < 4e-7 and o["shuffled_dF"] < 2e-6, o
        assert o["owned_dF"] < 2e-6 and o["owned_n"] >= 1490 and o["owned_coll"] == 1, o
        assert o["index_dF"] < 2e-6 and o["index_bytes"] == 4 * (3 * 3000 + 4 + 36), o   # forces + energy + virial parts
        assert o["batch_dE"] < 1e-9 and o["batch_dF_own"] < 2e-6, o
        # ANI-2dr on spatial shards: one all-to-all, a local system smaller than the box, the single-rank numbers
        assert o["d3_coll"] == 1 and o["d3_op"].startswith("all_to_all"), o
        assert o["d3_local"][1] == o["d3_local"][2] // 2 and o["d3_local"][0] < o["d3_local"][2], o
        assert o["d3_dE"] < 1e-7 * o["d3_local"][2] and o["d3_dF"] < 5e-6 * max(1.0, o["d3_Fmax"]) and o["d3_dW"] < 1e-4, o
        assert o["batch_bytes"] == 16 * o["batch_C"], o


def _worker_grads(rank, world, port, q):
    try:
        group = _setup(rank, world, port)
        import torch.distributed as dist
        from _util import load_golden, seeded_state
        from torchani_amd.models import ANI2x
        from torchani_amd.parallel import all_reduce_gradients

        dev = torch.device("cuda", 0)
        g = load_golden("rand_batch_ani2x")
        sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
        x = torch.from_numpy(g["coords"]).to(dev)
        C = sp.shape[0]
        target = torch.linspace(-1.0, 1.0, C, dtype=torch.float32, device=dev)

        def grads(lo, hi, reduce, flat=False):
            model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False)
            model.neural_networks.requires_grad_(True)
            model.set_enabled("energy_shifter", False)
            params = [p for p in model.neural_networks.parameters()]
            if flat:   # the gradients as views of the flat optimizer's one buffer: that buffer is the bucket
                from torchani_amd.optim import Adam
                from torchani_amd.parallel import _one_flat_group

                opt = Adam(params, lr=1e-4)
            e = model((sp[lo:hi], x[lo:hi])).energies
            loss = ((e.float() - target[lo:hi]) ** 2).sum()
            loss.backward()
            if flat:
                assert _one_flat_group(params) is opt._flat[0]
            if reduce:
                all_reduce_gradients(params, group=group)
            return torch.cat([p.grad.reshape(-1) for p in params])

        full = grads(0, C, False)
        lo, hi = rank * C // world, (rank + 1) * C // world
        part = grads(lo, hi, True)
        part_flat = grads(lo, hi, True, flat=True)
        scale = float(full.abs().max())
        q.put((rank, {"err": float((part - full).abs().max()), "err_flat": float((part_flat - full).abs().max()), "scale": scale}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback

        q.put((rank, {"error": repr(e) + "\n" + traceback.format_exc()}))


def test_two_ranks_all_reduce_gradients_equals_full_batch():
    res = _run(_worker_grads)
    for r, o in res.items():
        assert o["scale"] > 0 and o["err"] < 2e-5 * o["scale"] and o["err_flat"] < 2e-5 * o["scale"], o


def test_bench_two_ranks_gloo():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run), ranks sharing the GPU over gloo."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--waters-side", "16", "--dist-backend", "gloo", "--no-dense-stage", "--no-secondary", "--no-cpu-baseline", "--parity-sample", "128"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["steps"] == 2
    coll = r["collective"]
    assert coll["collectives_per_step"] == 1 and coll["world_size"] == 2
    assert coll["owned_atoms"] == r["config"]["n_atoms"] // 2 and coll["local_atoms"] <= r["config"]["n_atoms"]
    # what a rank sends: its halo rows to their owner + the partial energy (two fp32 words) to the other rank
    assert coll["bytes_per_step"] == 4 * (3 * coll["halo_atoms"] + 2)
    assert len(r["stages_ms_per_rank"]) == 2 and all("aev_forward" in s for s in r["stages_ms_per_rank"])
    seen = coll["ranks_seen"]
    assert sorted(s["rank"] for s in seen) == [0, 1] and len({s["pid"] for s in seen}) == 2
    assert seen[0]["peers"] == [1] and seen[1]["peers"] == [0]
    # the N > 1 line carries correctness evidence: rank 0's owned atoms against the fp64 oracle
    par = r["parity_sample"]
    assert par is not None and "rank 0 owns" in par["atoms_sampled_from"]
    assert par["max_dE_atom"] <= par["regression_gate_dE_atom"] and par["max_dF"] <= par["regression_gate_dF"]


def test_bench_eight_ranks_gloo_equals_one_rank():
    """First contact with 8 ranks, as far as one GPU goes: ``bench.py --gpus 8`` (eight processes sharing the GPU over gloo,
    a 41 472-atom box: slabs of 9.3 A against a reach of 6.1 A) -- every rank is seen, force rows travel to the two slab
    neighbours only, the bytes are what the plan says, rank 0's oracle sample of atoms it owns is inside the regression gates,
    and the total energy equals the single-rank run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--steps", "2", "--warmup", "2", "--waters-side", "24", "--no-dense-stage", "--no-secondary", "--no-cpu-baseline"]
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--parity-sample", "0"] + common,
                        capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p1.returncode == 0, p1.stdout[-2000:] + p1.stderr[-4000:]
    one = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    p8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo",
                         "--parity-sample", "128"] + common, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p8.returncode == 0, p8.stdout[-2000:] + p8.stderr[-4000:]
    lines = [ln for ln in p8.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p8.stdout[-2000:]
    r = json.loads(lines[0])
    n = r["config"]["n_atoms"]
    assert r["n_gpus"] == 8 and n == one["config"]["n_atoms"] == 3 * 24 ** 3
    coll = r["collective"]
    assert coll["world_size"] == 8 and coll["collectives_per_step"] == 1
    assert coll["transport"]["transport"] == "collective" and coll["transport"]["fell_back"] is None
    seen = sorted(coll["ranks_seen"], key=lambda s_: s_["rank"])
    assert [s_["rank"] for s_ in seen] == list(range(8)) and len({s_["pid"] for s_ in seen}) == 8
    assert sum(s_["owned_atoms"] for s_ in seen) == n
    for s_ in seen:
        assert sorted(s_["peers"]) == sorted({(s_["rank"] - 1) % 8, (s_["rank"] + 1) % 8}), s_
        halo = s_["local_atoms"] - s_["owned_atoms"]
        # halo force rows (3 fp32 words each) to their owners + the partial energy (two fp32 words) to the 7 other ranks
        assert s_["sent_bytes_per_step"] == 4 * (3 * halo + 2 * 7), s_
    par = r["parity_sample"]
    assert par is not None and par["n"] == 32 and "rank 0 owns" in par["atoms_sampled_from"]
    assert par["max_dE_atom"] <= par["regression_gate_dE_atom"] and par["max_dF"] <= par["regression_gate_dF"]
    assert abs(r["energy_Ha"] - one["energy_Ha"]) < 1e-7 * n


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` with no RANK / WORLD_SIZE in the environment -- the spelling the driver uses -- starts
    the two ranks itself (here over gloo on the one GPU) and prints the ranks' one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--waters-side", "16",
           "--dist-backend", "gloo", "--no-dense-stage", "--no-secondary", "--no-cpu-baseline", "--parity-sample", "128"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["value"] > 0
    coll = r["collective"]
    assert coll["world_size"] == 2 and coll["collectives_per_step"] == 1 and len(coll["ranks_seen"]) == 2
    assert "read one step late" in coll["validity_check"]


@pytest.mark.parametrize("transport", ["collective", "p2p"])
def test_rccl_calls_of_the_step_run_on_this_gpu(transport):
    """The collectives of the N > 1 step as RCCL executes them (backend "nccl", a process group of world size 1 with the
    collectives forced: tools/rccl_world1.py): the spatial exchange in fp32 and as int64 fixed-point words, the gather of
    owned rows and the index-range all-reduce reproduce the plain single-GPU energies, forces and virial -- through the
    one all_to_all_single and through the paired isend / irecv fallback."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TORCHANI_AMD_FORCE_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TORCHANI_AMD_EXCHANGE=transport)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "rccl_world1.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "world-1 collectives ok" in p.stdout and p.stdout.count("partition ") == 4, p.stdout[-2000:]
    assert f"transport {transport}" in p.stdout, p.stdout[-2000:]
