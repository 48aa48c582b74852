"""CPU baseline from the REFERENCE itself: torchani.grad.energies_and_forces (pyaev + cell_list, fp32) on the
same periodic water box generator bench.py uses, all host cores of THIS (build) container.

    python tools/ref_cpu_baseline.py --side 20 [--reps 5]      -> profiles/ref_cpu_baseline.json

The reference lives only in the build container (/root/reference cannot travel to the GPU box), so bench.py
reports this recorded number as ``cpu_baseline_reference`` next to the same-run ``cpu_baseline`` (the C oracle,
"port").  Weights: ANI-2x architecture, seeded random parameters (the published ones are a download).
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
    cpu = subprocess.run("lscpu | grep 'Model name' | sed 's/.*: *//'", shell=True, capture_output=True, text=True).stdout.strip()
    res = {
        "value": n / med, "unit": "atom*steps/s", "cores": cores, "kind": "reference", "cpu": cpu,
        "ms_per_step": med * 1e3, "n_atoms": n,
        "sample": f"{n}-atom periodic water box (bench.water_box({args.side})), torchani.grad.energies_and_forces, "
                  f"pyaev + cell_list, fp32, {cores} threads, warm-up {args.warmup}, median of {args.reps}",
        "where": "build container (the reference cannot travel to the GPU box); recorded, not measured in the bench run",
        "energy_Ha": float(e.detach()[0]), "max_abs_force": float(f.abs().max()),
    }
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
