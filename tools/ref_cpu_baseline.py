"""CPU baseline from the REFERENCE itself, calibrated against the oracle port: torchani.grad.energies_and_forces (pyaev +
cell_list, fp32; /root/reference/torchani/grad.py:263-290) AND oracle/ani_oracle.c (float build, OpenMP) on the same periodic
water box (bench.water_box), the same thread count, in the same process, on THIS (build) container's cores.

    python tools/ref_cpu_baseline.py --side 24 [--reps 5]      -> profiles/ref_cpu_baseline.json

The reference lives only in the build container (/root/reference cannot travel to the GPU box), so bench.py
reports the recorded reference number as ``cpu_baseline_reference`` next to the same-run ``cpu_baseline`` (the C oracle,
"port"), and -- from the ``port_over_reference`` ratio measured here -- ``cpu_baseline.reference_equivalent`` = what the reference
would do on the GPU box's host if the ratio carries over.  Where /root/reference IS importable, bench.py times the reference
itself (bench.reference_cpu_baseline).  Weights: ANI-2x architecture, seeded random parameters (the published ones are a
download).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def import_reference():
    os.environ["TORCHANI_NO_WARN_EXTENSIONS"] = "1"

    class _Any:
        def __class_getitem__(cls, k):
            return cls

    for name in ("h5py", "zarr"):
        m = types.ModuleType(name)
        for k in ("File", "Group", "Dataset", "Datatype"):
            setattr(m, k, type(k, (_Any,), {}))
        sys.modules[name] = m
    sys.path.insert(0, "/root/reference")
    import torchani  # noqa: F401

    return torchani


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=20, help="waters per box edge (bench.water_box)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "ref_cpu_baseline.json"))
    args = ap.parse_args()
    import torch

    from bench import water_box

    torchani = import_reference()
    from torchani.arch import Assembler
    from torchani.grad import energies_and_forces
    from torchani.utils import SYMBOLS_2X

    cores = os.cpu_count()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    asm = Assembler()
    asm.set_symbols(SYMBOLS_2X)
    asm.set_global_cutoff_fn("cosine")
    asm.set_aev_computer(radial="ani2x", angular="ani2x", strategy="pyaev")
    asm.set_atomic_networks(ctor="ani2x")
    asm.set_neighborlist("cell_list")
    asm.set_gsaes_as_self_energies("wb97x-631gd")
    model = asm.assemble(8)
    model.requires_grad_(False)
    sp, x, cell = water_box(args.side)
    n = sp.shape[1]
    znum = torch.tensor([1, 6, 7, 8, 16, 9, 17])[torch.from_numpy(sp)]   # element index -> atomic number
    coords = torch.from_numpy(x)
    cell_t = torch.from_numpy(cell)
    pbc = torch.tensor([True, True, True])
    times = []
    for k in range(args.warmup + args.reps):
        t0 = time.perf_counter()
        e, f = energies_and_forces(model, znum, coords, cell_t, pbc)
        dt = time.perf_counter() - t0
        if k >= args.warmup:
            times.append(dt)
        print(f"step {k}: {dt:.2f} s", flush=True)
    med = statistics.median(times)
    # the oracle port on the same box, the same threads, the same process
    import numpy as np

    from oracle import oracle as orc
    from torchani_amd.weights import arch_spec, random_state_dict

    os.environ["OMP_NUM_THREADS"] = str(cores)
    sd = random_state_dict("ani2x", 8, 0)
    symbols, _, _ = arch_spec("ani2x")
    dims, flat = orc.pack_networks(sd, symbols, 8)
    o32 = orc.Oracle("f32")
    sae = sd["energy_shifter.self_energies"].astype(np.float64)
    ptimes = []
    for k in range(1 + args.reps):
        t0 = time.perf_counter()
        o32.energy_forces(orc.params_2x(), sp, x, dims, flat, 8, sae=sae, cell=cell, pbc=(True, True, True), cell_list=True)
        dt = time.perf_counter() - t0
        if k >= 1:
            ptimes.append(dt)
        print(f"port step {k}: {dt:.2f} s ({o32.num_threads()} threads)", flush=True)
    pmed = statistics.median(ptimes)
    cpu = subprocess.run("lscpu | grep 'Model name' | sed 's/.*: *//'", shell=True, capture_output=True, text=True).stdout.strip()
    res = {
        "value": n / med, "unit": "atom*steps/s", "cores": cores, "kind": "reference", "cpu": cpu,
        "ms_per_step": med * 1e3, "n_atoms": n,
        "sample": f"{n}-atom periodic water box (bench.water_box({args.side})), torchani.grad.energies_and_forces, "
                  f"pyaev + cell_list, fp32, {cores} threads, warm-up {args.warmup}, median of {args.reps}",
        "where": "build container (the reference cannot travel to the GPU box); recorded, not measured in the bench run",
        "energy_Ha": float(e.detach()[0]), "max_abs_force": float(f.abs().max()),
        "port": {"value": n / pmed, "unit": "atom*steps/s", "cores": o32.num_threads(), "ms_per_step": pmed * 1e3,
                 "sample": f"oracle/ani_oracle.c float build, OpenMP, same box, same process, median of {args.reps}"},
        "port_over_reference": (n / pmed) / (n / med),
    }
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
