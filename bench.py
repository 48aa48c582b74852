"""Headline benchmark: ANI-2x energy+forces throughput (atom*steps/s) on a periodic water box.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is
one of the ranks; started plainly with --gpus N it launches the N ranks itself (torch.distributed.run on 127.0.0.1) and
passes their JSON line through.

Workload (BASELINE.json north_star / configs[3]): 8-member ANI-2x ensemble on a 2.3 M-atom periodic water
box (0.1 atoms/A^3), energies + forces.  It fits one MI355X, so N=1 runs the whole box and N>1 shards
the SAME box over the ranks (strong scaling): spatial slabs of the cell-sorted order, every rank works on its slab + halo,
ONE all-to-all per step over RCCL (halo force rows to the slab neighbours, the fp64 partial energies to everybody).
Weights are seeded random parameters of the ANI-2x architecture (the published ones are a download), data is synthetic.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      : the fused radial+angular AEV forward kernel against the HBM roofline
  roofline_mfma : the ensemble stack (fwd + input-gradient bwd): the fp16 MFMA flops issued by the split-fp16
                  path for the EXECUTED work (layer 0 skips all-zero AEV slabs) against the dense fp16 MFMA peak
  cpu_baseline  : the CPU oracle (float build, all host cores) timed on a bounded sub-box, N=1 only
  stages_ms     : per-stage device time of one step on this rank (HIP events on the engine's stream)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak (the f16x3 path issues 3 fp16 MFMA flops per fp32 flop)


def water_box(n_side: int, seed: int = 4, spacing: float = 3.107):
    """n_side^3 TIP3P-geometry waters on a jittered lattice, cubic periodic box at 0.1 atoms/A^3.
    Element indices: O=3, H=0 (ANI-2x order H C N O S F Cl)."""
    rs = np.random.RandomState(seed)
    L = n_side * spacing
    ax = (np.arange(n_side, dtype=np.float32) + 0.5) * spacing
    o = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    n = o.shape[0]
    o = o + rs.uniform(-0.35, 0.35, (n, 3)).astype(np.float32)

    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    a = unit(rs.normal(size=(n, 3)).astype(np.float32))
    b = unit(np.cross(a, rs.normal(size=(n, 3)).astype(np.float32)))
    th = np.deg2rad(104.52) / 2
    h1 = o + 0.9572 * (np.cos(th) * a + np.sin(th) * b)
    h2 = o + 0.9572 * (np.cos(th) * a - np.sin(th) * b)
    x = np.stack([o, h1, h2], 1).reshape(1, 3 * n, 3).astype(np.float32)
    sp = np.tile(np.array([3, 0, 0], dtype=np.int64), n).reshape(1, 3 * n)
    cell = (np.eye(3) * L).astype(np.float32)
    return sp, x, cell


def mlp_flops_per_atom(species_idx: np.ndarray, l0_cols: float = None) -> float:
    """fwd + input-gradient bwd flops of the 8-member ensemble, averaged over the atoms.
    l0_cols: AEV columns layer 0 actually multiplies (slab skipping); default = all of them."""
    from torchani_amd.constants import HIDDEN_DIMS_2X, SYMBOLS_2X, aev_constants_2x

    L = aev_constants_2x().out_dim
    per = []
    for s in SYMBOLS_2X:
        d = (L if l0_cols is None else l0_cols,) + tuple(HIDDEN_DIMS_2X[s]) + (1,)
        per.append(8 * sum(2 * d[i] * d[i + 1] for i in range(len(d) - 1)))
    per = np.asarray(per, dtype=np.float64)
    cnt = np.bincount(species_idx[species_idx >= 0], minlength=len(per))
    return float(2.0 * (per * cnt).sum() / cnt.sum())


def time_stage(fn, reps):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def cpu_model() -> str:
    """Model name and core count of the host CPU (lscpu's "Model name", from /proc/cpuinfo)."""
    try:
        with open("/proc/cpuinfo") as fh:
            names = [ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")]
        return f"{names[0]} ({len(names)} logical CPUs)" if names else "unknown"
    except OSError:
        return "unknown"


def cpu_baseline(n_side: int, seed: int):
    """CPU oracle ('port' of the reference path, float build, OpenMP over all host cores) on a bounded
    periodic sub-box of the same density; returns the cpu_baseline object."""
    from oracle import oracle as orc
    from torchani_amd.weights import arch_spec, random_state_dict

    sp, x, cell = water_box(n_side, seed=seed)
    sd = random_state_dict("ani2x", 8, 0)
    symbols, _, _ = arch_spec("ani2x")
    dims, flat = orc.pack_networks(sd, symbols, 8)
    o32 = orc.Oracle("f32")
    cores = o32.num_threads()
    # warm the library once (page-in, OpenMP thread team) on a small box before the timed evaluation
    sp_w, x_w, cell_w = water_box(8, seed=seed + 1)
    o32.energy_forces(orc.params_2x(), sp_w, x_w, dims, flat, 8, sae=sd["energy_shifter.self_energies"].astype(np.float64),
                      cell=cell_w, pbc=(True, True, True), cell_list=True)
    t0 = time.perf_counter()
    o32.energy_forces(orc.params_2x(), sp, x, dims, flat, 8, sae=sd["energy_shifter.self_energies"].astype(np.float64),
                      cell=cell, pbc=(True, True, True), cell_list=True)
    dt = time.perf_counter() - t0
    n = sp.shape[1]
    return {
        "value": n / dt, "unit": "atom*steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "sample": f"{n}-atom periodic water sub-box (same density and model), 1 energy+forces step, "
                  f"{dt:.1f} s, oracle/ani_oracle.c float build with OpenMP",
    }


def rccl_transport_detail():
    """What RCCL's INFO log of THIS process says about its channels: {"P2P/IPC": n, "SHM": n, "NET/...": n, ...} from the
    "a[x] -> b[y] via <transport>" lines (NCCL_DEBUG_FILE set by main() for N > 1), or None if there is no log."""
    pat = os.environ.get("NCCL_DEBUG_FILE")
    if not pat:
        return None
    path = pat.replace("%p", str(os.getpid())).replace("%h", socket.gethostname())
    if not os.path.exists(path):
        return None
    counts = {}
    try:
        with open(path, errors="replace") as fh:
            for ln in fh:
                if " via " in ln and "->" in ln:
                    t = ln.rsplit(" via ", 1)[1].split()[0].strip()
                    counts[t] = counts.get(t, 0) + 1
    except OSError:
        return None
    return counts or {"log": "no channel lines"}


def reference_cpu_baseline(n_side: int = 12, reps: int = 2):
    """The REFERENCE itself (torchani.grad.energies_and_forces, pyaev + cell_list, fp32; /root/reference/torchani/grad.py:263-290)
    timed on this host's cores on a bounded periodic water box -- only where /root/reference is importable (the build
    container; the GPU box has no /root/reference, and nothing of it may travel).  None otherwise."""
    if not os.path.isdir("/root/reference/torchani"):
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from ref_cpu_baseline import import_reference

        import_reference()
        from torchani.arch import Assembler
        from torchani.grad import energies_and_forces
        from torchani.utils import SYMBOLS_2X
    except Exception as exc:   # noqa: BLE001  (an optional leg: the bench line says why it is missing)
        return {"error": f"reference not importable: {exc}"}
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    asm = Assembler()
    asm.set_symbols(SYMBOLS_2X)
    asm.set_global_cutoff_fn("cosine")
    asm.set_aev_computer(radial="ani2x", angular="ani2x", strategy="pyaev")
    asm.set_atomic_networks(ctor="ani2x")
    asm.set_neighborlist("cell_list")
    asm.set_gsaes_as_self_energies("wb97x-631gd")
    model = asm.assemble(8)
    model.requires_grad_(False)
    sp, x, cell = water_box(n_side, seed=5)
    znum = torch.tensor([1, 6, 7, 8, 16, 9, 17])[torch.from_numpy(sp)]
    times = []
    for _ in range(1 + reps):
        t0 = time.perf_counter()
        energies_and_forces(model, znum, torch.from_numpy(x), torch.from_numpy(cell), torch.tensor([True, True, True]))
        times.append(time.perf_counter() - t0)
    dt = min(times[1:])
    n = sp.shape[1]
    return {"value": n / dt, "unit": "atom*steps/s", "cores": cores, "kind": "reference", "cpu_model": cpu_model(),
            "sample": f"{n}-atom periodic water box, torchani.grad.energies_and_forces (pyaev + cell_list, fp32), best of {reps}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--waters-side", type=int, default=92, help="waters per box edge (92 -> 2,336,064 atoms)")
    ap.add_argument("--cpu-side", type=int, default=36, help="waters per edge of the CPU-baseline sub-box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-2 / config-3 secondary measurements")
    ap.add_argument("--parity-sample", type=int, default=512,
                    help="atoms of the headline box checked against the fp64 oracle after the timed loop (N = 1; 0 = skip)")
    ap.add_argument("--no-dense-stage", action="store_true",
                    help="skip the extra (untimed-region) dense-MLP stage timing, e.g. under rocprofv3 so the "
                         "kernel statistics hold the product configuration only")
    ap.add_argument("--dist-backend", default=None,
                    help="development aid: torch.distributed backend override (gloo lets several ranks share one "
                         "GPU to exercise the N>1 code path without RCCL); implies --share-gpu")
    ap.add_argument("--emulate-shard", default=None, metavar="RANK/WORLD",
                    help="development aid (N=1 only): time what rank RANK of WORLD does in a step, without "
                         "the collective -- prints ms/step and exits")
    ap.add_argument("--emulate-collective-bytes", action="store_true",
                    help="with --emulate-shard: also time the pack / unpack of the step's all-to-all with the real byte plan "
                         "of that rank (no wire: the only term a real N > 1 run adds is bytes / link rate)")
    ap.add_argument("--emulate-option-c", action="store_true",
                    help="with --emulate-shard: cost SURVEY 8(e) option C instead -- a REDUNDANT halo of twice the reach, the "
                         "rank evaluates its owned atoms AND the halo atoms within one reach of them, so every force on an owned "
                         "atom is complete locally and the only collective is an all-reduce of the scalar energy")
    ap.add_argument("--partition-skin", type=float, default=1.0,
                    help="N > 1: skin (Angstrom) of the spatial shards' halos; the partition is reused until an atom moved skin/2")
    ap.add_argument("--mlp-chunk", type=int, default=0, help="development aid: atoms per launch group of the network stage")
    ap.add_argument("--two-product-backward", action="store_true",
                    help="NOT the default arithmetic: the backward GEMMs of the network kernel with two products instead of "
                         "three (ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS); the line is labelled and its parity sample is held to "
                         "north_star's gates (1e-5 Ha, 1e-4 Ha/A) instead of this package's regression gates")
    ap.add_argument("--shuffle", action="store_true",
                    help="permute the atom order of the box (the spatial shards must not depend on it)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # started plainly: launch the ranks (one process per GPU) and hand their output through
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: RCCL between processes needs it on this driver)
        sys.exit(subprocess.call(cmd, env=env))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and args.dist_backend in (None, "nccl"):
        # which transport RCCL picks between the ranks (P2P/IPC over xGMI, SHM, NET/Socket) is in its INFO log only: every rank
        # writes its own file, parsed into collective.ranks_seen[*].rccl_transports after the timed loop (set before RCCL
        # initialises; a caller's own NCCL_DEBUG settings win)
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,P2P")
        os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/anihip_rccl_{os.getpid()}_%p.log")

    from torchani_amd import _lib
    from torchani_amd.models import ANI2x
    from torchani_amd.parallel import exchange_transport, init_from_env, shard_range

    rank, world, local, group = init_from_env(args.dist_backend)
    if args.dist_backend == "gloo":
        local = 0   # every rank on GPU 0
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.lib()

    sp_np, x_np, cell_np = water_box(args.waters_side)
    n_atoms = sp_np.shape[1]
    if args.shuffle:
        perm = np.random.RandomState(11).permutation(n_atoms)
        sp_np, x_np = np.ascontiguousarray(sp_np[:, perm]), np.ascontiguousarray(x_np[:, perm])
    species = torch.from_numpy(sp_np).to(dev)
    coords = torch.from_numpy(x_np).to(dev)
    cell = torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell", row_capacity=128)
    # Spatial shards are cut with a skin, as an MD driver would run them: halos 1 A wider, and the partition is kept until an
    # atom has moved 0.5 A (the coordinates of this bench are static, so it is cut once, in warmup; the cost of cutting it
    # is reported as collective.partition_ms)
    model.partition_skin = args.partition_skin
    model.two_product_backward = bool(args.two_product_backward)
    if args.mlp_chunk > 0:
        model.mlp_chunk = args.mlp_chunk

    def step():
        # (overflow is checked once after the timed loop instead of with a host sync per step).  N > 1: spatial shards,
        # every rank ends the step with the total energy and the forces of the atoms it owns (reduce_forces=False: no
        # gather of the other ranks' forces -- a domain-decomposed MD step does not need them)
        if group is not None:
            # an MD driver hands over NEW coordinates every step: bump the tensor's version so that the step pays what a
            # moving system pays -- the partition's validity flags (queued on the device, read one step late) -- instead of
            # hitting the "same tensor, same version" shortcut of a static input
            coords.add_(0.0)
        return model.energies_and_forces(species, coords, cell, pbc, group=group, check_overflow=False,
                                         reduce_forces=False)

    if args.emulate_shard:
        r, wd = (int(v) for v in args.emulate_shard.split("/"))
        for _ in range(2):
            model.energies_and_forces(species, coords, cell, pbc, shard=(r, wd), check_overflow=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.energies_and_forces(species, coords, cell, pbc, shard=(r, wd), check_overflow=False)
        torch.cuda.synchronize()
        lc = model.last_collective
        print(f"shard {r}/{wd}{' (shuffled input)' if args.shuffle else ''}: {(time.perf_counter() - t0) / args.steps * 1e3:.3f} "
              f"ms/step (no collective); local system {lc['n_local']} atoms = {lc['n_owned']} owned + {lc['n_halo']} halo")
        # the stages of that rank's step on its local system
        eng = model.aev_computer.engine()
        sp32 = species.to(torch.int32).contiguous()
        part = model._spatial_partition(sp32, coords, cell, pbc, r, wd)
        sp_e, order = model._engine_species(sp32)   # (the kernels' species numbering, models.ANI.compact_species)
        lo, hi, nl = part.n_left, part.n_left + part.n_owned, part.n_local
        if args.emulate_option_c:
            # option C: the local system reaches TWICE as far (everything an owned atom's force depends on), and the central
            # atoms are the owned ones plus the halo atoms within one reach of the slab -- the halo is sorted by layer along the
            # slab axis, so at uniform density those are the inner half of each side
            part = type(part)(coords, cell, pbc, wd, r, 2.0 * model._spatial_reach() + model.partition_skin, sp32,
                              skin=model.partition_skin)
            lo, hi, nl = part.n_left - part.n_left // 2, part.n_left + part.n_owned + part.n_right // 2, part.n_local
            print(f"  option C (redundant halo, scalar all-reduce only): local system {nl} atoms, central atoms {hi - lo} "
                  f"for {part.n_owned} owned")
        sp_l = part.local(sp_e).view(1, -1).contiguous()
        x_l = part.local(coords, 3).view(1, -1, 3).contiguous()
        packed = model.neural_networks._pack(dev, order)
        st = {f"partition (cut once per skin {model.partition_skin} A of motion)": time_stage(
            lambda: type(part)(coords, cell, pbc, wd, r, model._spatial_reach(), sp32, skin=model.partition_skin), 3)}
        st["gather local system"] = time_stage(lambda: (part.local(sp32), part.local(coords, 3)), 3)
        nbrs = eng.neighbors(sp_l, x_l, cell, pbc, lo=lo, hi=hi, mode="cell", row_cap=128)
        st["neighbors"] = time_stage(lambda: eng.neighbors(sp_l, x_l, cell, pbc, lo=lo, hi=hi, mode="cell", row_cap=128), 3)
        aev, mask = eng.forward_update(sp_l, nbrs)   # (kept rows, updated in place: the product path)
        st["aev_forward"] = time_stage(lambda: eng.forward_update(sp_l, nbrs), 3)
        ae = torch.zeros(nl, dtype=torch.float32, device=dev)
        gaev = torch.zeros_like(aev)
        st["mlp_fwd_bwd"] = time_stage(lambda: packed.forward_backward(sp_l, aev, lo=lo, hi=hi, atomic_e=ae, grad_aev=gaev,
                                                                       chunk=model.mlp_chunk, slab_mask=mask, shard_rows=True), 3)
        gc = torch.zeros((nl, 3), dtype=torch.float32, device=dev)
        st["aev_backward"] = time_stage(lambda: eng.backward(sp_l, nbrs, gaev, gc, shard_rows=True, slab_mask=mask), 3)
        st["scatter results"] = time_stage(lambda: (part.scatter_local(gc), part.scatter_owned(ae)), 3)
        if args.emulate_collective_bytes:
            # the step's ONE all-to-all with the real byte plan of this rank, without the wire: pack (gather of the halo rows
            # by owner + the tail) and unpack (index_add of the received rows, sum of the W tails) are timed; what a real run
            # adds is bytes / link rate
            from torchani_amd.parallel import EMULATE_WIRE

            tail = torch.zeros(10, dtype=torch.float64, device=dev)   # energy + nine virial words
            st["exchange pack+unpack (no wire)"] = time_stage(lambda: part.exchange(gc, tail, EMULATE_WIRE), 5)
            sent = {int(o): 12 * int(c) for o, c in enumerate(part.send_counts) if c and o != r}
            recv = {int(h): 12 * int(c) for h, c in enumerate(part.recv_counts) if c and h != r}
            print(f"  exchange byte plan of rank {r}/{wd}: force rows sent {sent} B, received {recv} B, tail 80 B to each of "
                  f"{wd - 1} ranks; bytes per step {part.last_bytes}")
        print("  stages ms: " + "  ".join(f"{k} {v:.3f}" for k, v in st.items()))
        per_step = sum(v for k, v in st.items() if not k.startswith("partition"))
        print(f"  sum of the per-step stages: {per_step:.3f} ms")
        return

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if group is not None:
        torch.distributed.barrier(group)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for k in range(args.steps):
        out = step()
        marks[k + 1].record()
    torch.cuda.synchronize()
    if group is not None:
        torch.distributed.barrier(group)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    elapsed_rank = elapsed
    if group is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    model.aev_computer.last_neighbors().raise_on_overflow()
    assert torch.isfinite(out.energies).all() and torch.isfinite(out.forces).all()
    # ---- is the headline result RIGHT?  Sampled atoms of the full box against the fp64 oracle (outside the timed region):
    # the cluster within 2 Rcr of an atom reproduces its energy and force in the periodic box exactly (oracle/sampled_parity.py)
    # N > 1: rank 0 checks atoms it OWNS (forces stay with their owners; the clusters are cut from the whole box, which every
    # rank holds), a quarter of the sample so that the other ranks do not wait long at the final barrier
    parity = None
    if rank == 0 and args.parity_sample > 0:
        from oracle.sampled_parity import sampled_parity

        sd_np = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        owned = None
        if group is not None:
            owned = model.__dict__["_spatial_cache"][1].owned_idx.cpu().numpy()   # (the partition of the last step)
        parity = sampled_parity(species, coords, cell, out.atomic_energies, out.forces, sd_np, "ani2x", 8,
                                n_sample=args.parity_sample if group is None else max(32, args.parity_sample // 4), seed=7,
                                candidates=owned,
                                # (a rank started by torch.distributed.run inherits OMP_NUM_THREADS=1; the other ranks wait at
                                # the final barrier meanwhile: the oracle gets the host's cores but two per rank)
                                threads=None if group is None else max(1, (os.cpu_count() or 1) - 2 * world))
        parity["atoms_sampled_from"] = "the whole box" if group is None else f"the {len(owned)} atoms rank 0 owns"
        gates = ("gate_dE_atom", "gate_dF") if args.two_product_backward else ("regression_gate_dE_atom", "regression_gate_dF")
        assert parity["max_dE_atom"] <= parity[gates[0]] and parity["max_dF"] <= parity[gates[1]], \
            f"headline result disagrees with the oracle: {parity}"

    # ---- per-stage device timing on this rank's shard (outside the timed region) -----------------------
    eng = model.aev_computer.engine()
    sp32 = species.to(torch.int32).contiguous()
    sp_given = sp32
    sp32, sp_order = model._engine_species(sp32)   # (the kernels' species numbering, models.ANI.compact_species)
    lo, hi = 0, n_atoms
    n_local = n_atoms
    coords_l = coords
    sorted_copy = group is None and model._wants_locality_sort(sp_given, coords, cell, pbc, species)
    if group is not None or sorted_copy:   # this rank's local system [left halo | owned | right halo] of the spatial
        # shards (one GPU, incoherent atom order: the cell-sorted copy the step works on, models.ANI.locality_sort)
        part = model._spatial_partition(sp_given, coords, cell, pbc, rank, world)
        sp32 = part.local(sp32).view(1, -1).contiguous()
        sp_given = part.local(sp_given).view(1, -1).contiguous()
        coords_l = part.local(coords, 3).view(1, -1, 3).contiguous()
        lo, hi, n_local = part.n_left, part.n_left + part.n_owned, part.n_local
    packed = model.neural_networks._pack(dev, sp_order)
    st = {}
    reps = 3
    nbrs = eng.neighbors(sp32, coords_l, cell, pbc, lo=lo, hi=hi, mode="cell", row_cap=128)
    st["neighbors"] = time_stage(lambda: eng.neighbors(sp32, coords_l, cell, pbc, lo=lo, hi=hi, mode="cell", row_cap=128), reps)
    # rows [hi - lo, L] and per-atom slab flags in the engine's kept buffers, updated in place: the product path
    aev, mask = eng.forward_update(sp32, nbrs)
    st["aev_forward"] = time_stage(lambda: eng.forward_update(sp32, nbrs), reps)
    if not args.no_dense_stage:   # (for comparison: every row written in full into a buffer of the caller, as AEVComputer.forward does)
        full = torch.empty_like(aev)
        st["aev_forward_full_rows"] = time_stage(lambda: eng.forward(sp32, nbrs, out=full, shard_rows=True), reps)
        del full
    ae = torch.zeros(n_local, dtype=torch.float32, device=dev)
    gaev = torch.zeros_like(aev)
    # (the flags the step itself hands to the network stage: models.ANI._tile_hint -- one launch per species with compile-time
    # network widths where every species present has many rounds of tiles -- + the opt-in two-product backward)
    stage_hint = model._tile_hint(sp_given, sp32, hi - lo) | (_lib.MLP_FLAG_BWD_TWO_PRODUCTS if args.two_product_backward else 0)
    st["mlp_fwd_bwd"] = time_stage(
        lambda: packed.forward_backward(sp32, aev, lo=lo, hi=hi, atomic_e=ae, grad_aev=gaev, chunk=model.mlp_chunk,
                                        slab_mask=mask, shard_rows=True,
                                        tile_hint=stage_hint), reps)
    if not args.no_dense_stage:
        st["mlp_fwd_bwd_dense"] = time_stage(
            lambda: packed.forward_backward(sp32, aev, lo=lo, hi=hi, atomic_e=ae, grad_aev=gaev,
                                            chunk=model.mlp_chunk, shard_rows=True), 2)
    gc = torch.zeros((n_local, 3), dtype=torch.float32, device=dev)
    st["aev_backward"] = time_stage(lambda: eng.backward(sp32, nbrs, gaev, gc, shard_rows=True, slab_mask=mask), reps)
    meta = nbrs.meta[lo:hi, 1].to(torch.int64) & 0xFFFFFFFF
    n_a = float((meta & 0xFFFF).double().mean())
    n_r = n_a + float((meta >> 16).double().mean())
    n_shard = hi - lo
    # algorithmic bytes per atom of the fused AEV forward: angular 3584 + 20 n_a + 8, radial 448 + 8 n_r
    bytes_per_atom = 3584 + 20 * n_a + 8 + 448 + 8 * n_r
    aev_gbs = bytes_per_atom * n_shard / (st["aev_forward"] * 1e-3) / 1e9
    # backward: read dE/dAEV 3584 + 448, the row 20 n_a + 8 n_r, write 12 (own force) + 12 n_r (pushes)
    bytes_per_atom_bwd = 3584 + 448 + 20 * n_a + 8 * n_r + 12 + 12 * n_r
    bwd_gbs = bytes_per_atom_bwd * n_shard / (st["aev_backward"] * 1e-3) / 1e9
    # HBM traffic of the AEV forward kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB per
    # dispatch; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md) measured on a smaller box of
    # the same density and committed under profiles/; scaled by the atom count of this launch
    aev_traffic = aev_full_traffic = bwd_traffic = mlp_traffic = nbr_traffic = None
    pmc_file = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc.json", "r05_pmc.json", "r04_pmc_l0b.json", "r03_pmc_aev.json"))
                     if os.path.exists(f)), "")
    # SQ counters of the same kernels (rocprofv3 --pmc, tools/gpu_r6_profile.sh): what bounds each kernel, as counters
    sq = {}
    sq_file = os.path.join(ROOT, "profiles", "r06_pmc_sq.json")
    if os.path.exists(sq_file):
        with open(sq_file) as fh:
            sq = json.load(fh).get("kernels", {})

    def sq_of(kernel, *keys):
        e = sq.get(kernel, {})
        return {k: e[k] for k in keys if k in e} or None
    if os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            pm = json.load(fh)
        per_atom = {k: (v["fetch_size_kb"] * pm["fetch_correction"] + v["write_size_kb"]) * 1024.0 / pm["n_atoms"]
                    for k, v in pm["kernels"].items()}
        aev_traffic, bwd_traffic = per_atom["k_aev_fwd3"] * n_shard, per_atom["k_aev_bwd"] * n_shard
        # (the forward kernel's two roles, told apart in the counter file: the launch that writes every row -- what `roofline`
        # prices -- and the kept-rows update the timed step runs)
        fw = pm["kernels"]["k_aev_fwd3"]
        role = lambda r: ((fw[r]["fetch_size_kb"] * pm["fetch_correction"] + fw[r]["write_size_kb"]) * 1024.0 / pm["n_atoms"]   # noqa: E731
                          * n_shard) if r in fw else None
        aev_full_traffic = role("full_rows")
        if role("kept_rows") is not None:
            aev_traffic = role("kept_rows")
        nbr_traffic = per_atom["k_nbr_cell2"] * n_shard if "k_nbr_cell2" in per_atom else None
        # (the network stage: every kernel of it that the counter file knows, per atom of a launch)
        # (the counter file holds means per DISPATCH; the network kernels run once per launch group: dispatches per step =
        # their dispatch count over that of a kernel that runs once per step)
        once = max(1, pm["kernels"]["k_aev_bwd"].get("dispatches_FETCH_SIZE", 1))
        mlp_traffic = sum(per_atom.get(k, 0.0) * pm["kernels"][k].get("dispatches_FETCH_SIZE", once) / once
                          for k in ("k_mlp_fused", "k_gemm_l0b", "k_gemm_h2") if k in pm["kernels"]) * n_shard
    # layer 0 multiplies only the 32-column AEV slabs flagged for an atom (absent neighbor species give
    # identically-zero blocks): executed flops = dense flops with the AEV length replaced by the staged columns
    pop = mask[lo:hi].to(torch.int64) & 0xFFFFFFFF
    mean_slabs = float(sum(((pop >> b) & 1).double().mean() for b in range(32)))
    sp_owned = sp_given.reshape(-1)[lo:hi].cpu().numpy()
    flops_dense = mlp_flops_per_atom(sp_owned)
    flops_atom = mlp_flops_per_atom(sp_owned, l0_cols=32.0 * mean_slabs)
    mlp_tflops = flops_atom * n_shard / (st["mlp_fwd_bwd"] * 1e-3) / 1e12
    mlp_tflops_dense = (flops_dense * n_shard / (st["mlp_fwd_bwd_dense"] * 1e-3) / 1e12
                        if "mlp_fwd_bwd_dense" in st else None)

    res = {
        "metric": "atom*steps/sec (energy+forces) ANI-2x",
        "value": n_atoms * args.steps / elapsed,
        "unit": "atom*steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "arithmetic": "fp32 throughout; ensemble GEMMs as split-fp16 x3 MFMA with fp32 accumulation "
                          "(fp32-class accuracy: per-atom energies within 1e-7 Ha of the fp64 reference)",
            "workload": f"ANI-2x 8-member ensemble, {n_atoms}-atom periodic water box (0.1 atoms/A^3), "
                        "energy+forces, seeded random weights",
            "n_atoms": n_atoms, "box_A": float(cell_np[0, 0]),
            "sharding": "spatial slabs of the coordinate-sorted order (any input order), each rank works on its slab + a "
                        "5.1 A halo; ONE all-to-all per step: halo force rows to the slab neighbours only, partial "
                        "energies to every rank; forces stay with the rank that owns the atoms (not gathered); the "
                        "partition's validity is checked every step on the device and read one step late (no host sync)",
            "shuffled_input": bool(args.shuffle), "evaluated_on_cell_sorted_copy": bool(sorted_copy),
            # (False: the default arithmetic.  True: NOT the default -- see --two-product-backward)
            "two_product_backward": bool(args.two_product_backward),
        },
        "ms_per_step_median": median_ms,
        "roofline": {
            # `achieved` / `frac`: the kernel that MOVES the algorithmic bytes of SURVEY 8(d) -- every row written in full into a
            # caller's buffer (AEVComputer.forward; bytes moved = bytes counted) -- timed live in this run.  The step itself runs
            # the `kept_rows` variant: the engine keeps the rows between steps and the kernel rewrites only the slabs that were
            # or are flagged (anihip_aev_forward_update); it moves far fewer bytes (`bytes_moved`, from the committed counter
            # file), so its rate on the algorithmic bytes would be throughput on bytes it does not move and is not `frac`.
            # Neither variant is bound by HBM: `valu` holds the SQ counters (the kernel is bound by VALU issue).
            "kernel": "k_aev_fwd3<8,4,rec> (fused radial+angular AEV forward) writing all 1008 columns of every row",
            "bound": "hbm",
            "achieved": bytes_per_atom * n_shard / (st.get("aev_forward_full_rows", st["aev_forward"]) * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_per_atom * n_shard / (st.get("aev_forward_full_rows", st["aev_forward"]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "avg_launch_ms": st.get("aev_forward_full_rows", st["aev_forward"]),
            "traffic": aev_full_traffic, "algorithmic_bytes_per_atom": bytes_per_atom,
            "kept_rows": {"kernel": "k_aev_fwd3<8,4,rec,UPDATE>: the variant the timed step runs (rows kept by the engine and "
                                    "updated in place: only flagged 32-column slabs are rewritten)",
                          "avg_launch_ms": st["aev_forward"], "bytes_moved": aev_traffic,
                          "frac_on_moved": (aev_traffic / (st["aev_forward"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if aev_traffic else None,
                          "algorithmic_throughput_GBps": aev_gbs},
            "valu": sq_of("k_aev_fwd3", "valu_active_over_wave_cycles", "waves_per_simd_resident", "simd_valu_busy",
                          "wait_any_over_wave_cycles", "sq_insts_valu_per_atom", "sq_insts_salu_per_atom", "sq_insts_lds_per_atom"),
            "mean_radial_neighbors": n_r, "mean_angular_neighbors": n_a,
        },
        "roofline_bwd": {
            "kernel": "k_aev_bwd<8,4> (analytic AEV backward: radial by symmetric gather, angular pair loop)",
            "bound": "hbm", "achieved": bwd_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bwd_gbs / HBM_PEAK_GBS,
            "traffic": bwd_traffic, "algorithmic_bytes_per_atom": bytes_per_atom_bwd,
            "achieved_counter_GBps": (bwd_traffic / (st["aev_backward"] * 1e-3) / 1e9) if bwd_traffic else None,
            "avg_launch_ms": st["aev_backward"],
            "valu": sq_of("k_aev_bwd", "valu_active_over_wave_cycles", "waves_per_simd_resident", "simd_valu_busy",
                          "wait_any_over_wave_cycles", "sq_insts_valu_per_atom", "sq_insts_salu_per_atom", "sq_insts_lds_per_atom",
                          "sq_insts_vmem_wr_per_atom"),
        },
        "roofline_nbr": {
            # neighbor rows (binning + k_nbr_cell2 + finish): per central atom 16 B of packed position in, 24 B of row
            # metadata and 16 B per radial neighbor out; the launch duration is the whole STAGE (events
            # around anihip_nbr_build_cell), of which k_nbr_cell2 is ~80 % (profiles/: kernel statistics of the same command)
            "kernel": "neighbor stage: k_bin_* + k_nbr_cell2 (one wave per bin, candidates staged in LDS)", "bound": "hbm",
            "algorithmic_bytes_per_atom": 16.0 + 24.0 + 16.0 * n_r,
            "achieved": (16.0 + 24.0 + 16.0 * n_r) * n_shard / (st["neighbors"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": (16.0 + 24.0 + 16.0 * n_r) * n_shard / (st["neighbors"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": nbr_traffic,
            "achieved_counter_GBps": (nbr_traffic / (st["neighbors"] * 1e-3) / 1e9) if nbr_traffic else None,
            "avg_launch_ms": st["neighbors"],
        },
        "roofline_mfma": {
            "kernel": "ensemble fwd + input-gradient bwd: k_mlp_fused<2,1,CELU,L0B,H1,H2,H3> on 16x16x32 MFMA tiles, one launch per "
                      "species with compile-time network widths (layer 0 over flagged AEV slabs + hidden "
                      "stack + backward + layer-0 backward as phase 5: a workgroup owns a tile through all members), "
                      f"precision {packed.precision}",
            "traffic": mlp_traffic,   # HBM bytes of the stage per step from the committed counter file (round 3: 51.8e9)
            "achieved_counter_GBps": (mlp_traffic / (st["mlp_fwd_bwd"] * 1e-3) / 1e9) if mlp_traffic else None,
            "bound": "mfma",
            # the split-fp16 path issues 3 fp16 MFMA flops per fp32 flop it replaces: price the ISSUED fp16
            # flops of the EXECUTED (slab-skipped) work against the dense fp16 MFMA peak
            "achieved": 3.0 * mlp_tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": 3.0 * mlp_tflops / MFMA_F16_PEAK_TFLOPS,
            "fp32_equivalent_tflops": mlp_tflops, "fp32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
            "flops_per_atom_executed": flops_atom, "flops_per_atom_dense": flops_dense,
            "mean_active_slabs": mean_slabs,
            "dense_fp32_equivalent_tflops": mlp_tflops_dense,   # all 32 slabs multiplied (no masks)
            # SQ counters: matrix-pipe busy cycles and VALU issue per SIMD-cycle of the launch (profiles/r06_pmc_sq.json)
            "pipes": sq_of("k_mlp_fused", "simd_mfma_busy", "simd_valu_busy", "valu_active_over_wave_cycles",
                           "wait_any_over_wave_cycles", "lds_active_over_wave_cycles"),
            # the package runs this kernel at its power limit (rocm-smi: 1.29-1.37 kW at ~2.2 GHz): tools/mfma_power.hip,
            # profiles/r06_mfma_power.txt -- what a pure MFMA stream sustains on random fp16 operands
            "power_limited_mfma_peak_TFLOPS": {"v_mfma_f32_32x32x16_f16": 1292.0, "v_mfma_f32_16x16x32_f16": 1871.0,
                                               "zeros_32x32x16": 2449.0},
        },
        "stages_ms": st,
        "parity_sample": parity,
        "energy_Ha": float(out.energies.reshape(-1)[0]),   # (total energy of the box: the same number at every N)
    }
    if group is not None:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, st, group=group)
        lc = model.last_collective
        backend = torch.distributed.get_backend(group)
        res["stages_ms_per_rank"] = per_rank
        seen = [None] * world
        props = torch.cuda.get_device_properties(dev)
        # first contact with a multi-GPU node: can this rank's device reach the devices of its two slab neighbours directly
        # (hipDeviceCanAccessPeer), and what did RCCL choose per channel
        peer_access = {}
        for pr_ in part.peers:
            pl = pr_ if args.dist_backend != "gloo" else 0   # (one rank per GPU: local rank = rank on one node)
            try:
                peer_access[int(pr_)] = bool(pl == dev.index or torch.cuda.can_device_access_peer(dev.index, pl))
            except Exception as exc:   # noqa: BLE001
                peer_access[int(pr_)] = f"error: {exc}"
        torch.distributed.all_gather_object(seen, {
            "rank": rank, "local_rank": local, "device": f"cuda:{dev.index}", "name": props.name, "pid": os.getpid(),
            "peer_access": peer_access, "rccl_transports": rccl_transport_detail(),
            "peers": list(part.peers), "local_atoms": part.n_local, "owned_atoms": part.n_owned,
            "sent_bytes_per_step": lc["bytes"], "ms_per_step_this_rank": elapsed_rank / args.steps * 1e3}, group=group)
        part_ms = time_stage(lambda: type(part)(coords, cell, pbc, world, rank, model._spatial_reach(),
                                                species.to(torch.int32).view(-1), skin=model.partition_skin), 3)
        res["collective"] = {
            "collectives_per_step": lc["collectives_per_step"], "world_size": lc["world_size"],
            "bytes_per_step": lc["bytes"], "op": lc.get("op", "all_reduce(sum, fp32)"), "backend": backend,
            "ranks_seen": seen, "forces": "left with the owning rank (reduce_forces=False), not gathered",
            # "collective": ONE all_to_all_single with uneven pieces; "p2p": the same pieces as batched isend / irecv -- the
            # fallback every rank takes if the collective's first call raises (fell_back = its error message)
            "transport": exchange_transport(),
            "validity_check": "every timed step: device flags, read one step late (coords.add_(0.0) bumps the tensor version)",
            "local_atoms": lc.get("n_local"), "owned_atoms": lc.get("n_owned"), "halo_atoms": lc.get("n_halo"),
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
            # cutting the partition (sort + halo plan) is NOT part of every step: it is reused until an atom has moved
            # skin / 2 (models.ANI.partition_skin); the timed steps run on halos that are skin wider for it
            "partition_ms": part_ms, "partition_skin_A": model.partition_skin,
            "partition_reuse": "until an atom has moved skin/2; static coordinates here, so cut once in warmup",
        }
    if rank == 0 and world == 1 and not args.no_secondary:
        # BASELINE configs 2 / 3 on the parity fixtures' inputs (outside the timed headline region)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs

        torch.cuda.empty_cache()
        res["secondary"] = bench_configs.measure(dev)
        res["secondary"]["config5"] = bench_configs.measure_config5()
        if not args.two_product_backward and args.parity_sample > 0:
            # NOT the default arithmetic, reported beside it: the same box with the backward GEMMs of the network kernel on two
            # products (ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS, model.two_product_backward)
            from oracle.sampled_parity import sampled_parity

            model.two_product_backward = True
            for _ in range(3):
                out2 = step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(10):
                out2 = step()
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t2) / 10 * 1e3
            model.two_product_backward = False
            par2 = sampled_parity(species, coords, cell, out2.atomic_energies, out2.forces, sd_np, "ani2x", 8, n_sample=128, seed=7)
            res["secondary"]["two_product_backward"] = {
                "default": False, "ms_per_step": ms2, "atom_steps_per_s": n_atoms / ms2 * 1e3,
                "max_dE_atom": par2["max_dE_atom"], "max_dF": par2["max_dF"], "n_sample": par2["n"],
                "note": "opt-in: forces within north_star's 1e-4 Ha/A, outside this package's 5e-6 Ha/A regression gate"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_side, seed=5)
            ref_now = reference_cpu_baseline()
            if ref_now is not None:   # (/root/reference is importable: the build container, never the GPU box)
                res["cpu_baseline_reference_same_run"] = ref_now
            ref_file = os.path.join(ROOT, "profiles", "ref_cpu_baseline.json")
            if os.path.exists(ref_file):   # the reference itself (torchani.grad.energies_and_forces), recorded by
                with open(ref_file) as fh:   # tools/ref_cpu_baseline.py in the build container (it cannot travel)
                    res["cpu_baseline_reference"] = json.load(fh)
                ratio = res["cpu_baseline_reference"].get("port_over_reference")
                if ratio:
                    # the oracle port timed in THIS run on this host, divided by (port / reference) measured on one host,
                    # one thread count, one process in the build container (tools/ref_cpu_baseline.py): what the reference
                    # itself would do here if the ratio carries over
                    res["cpu_baseline"]["port_over_reference"] = ratio
                    res["cpu_baseline"]["reference_equivalent"] = res["cpu_baseline"]["value"] / ratio
        print(json.dumps(res))
    if group is not None:
        torch.distributed.barrier(group)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
