"""Throughput of the secondary BASELINE.json configurations on one GPU, on the SAME inputs as the parity fixtures
(tests/golden/gen_golden_configs.py):   python tools/bench_configs.py [--json]
  config 2: 256 molecules (frames of dataset/xyz_files/13.xyz + 28.xyz padded to A = 28; 5248 real atoms), batch mode,
            eager and as a HIP graph replay
  config 3: 1hz5 solvated to 46 357 atoms in a periodic box, cell mode; 1C17.pdb (16 649 atoms, no PBC)
bench.py imports ``measure()`` for the "secondary" object of its JSON line (outside the timed headline region).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def timeit(fn, warm=5, reps=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def measure(dev=None, reps2=50, reps3=20):
    from torchani_amd.models import ANI2x

    dev = dev or torch.device("cuda:0")
    out = {}
    with np.load(os.path.join(GOLD, "cfg2_xyz13_28_ani2x.npz")) as z:
        sp, x = z["species"].astype(np.int64), z["coords"]
    n_real = int((sp >= 0).sum())
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="batch")
    spd, xd = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
    model.auto_graph_atoms = 0
    dt = timeit(lambda: model.energies_and_forces(spd, xd, check_overflow=False), reps=reps2)
    f = model.graphed(spd, xd)
    dtg = timeit(lambda: f(xd), reps=reps2)
    model.auto_graph_atoms = 24000   # the default (models.ANI): energies_and_forces itself replays a graph from the third call on
    dta = timeit(lambda: model.energies_and_forces(spd, xd), reps=reps2)
    out["config2"] = {"workload": f"256 molecules (13.xyz / 28.xyz frames 0-127, A = 28, {n_real} real atoms), batch mode",
                      "ms_eager": dt * 1e3, "ms_graph_replay": dtg * 1e3, "ms_default_api": dta * 1e3,
                      "atom_steps_per_s_graph": n_real / dtg}
    if os.environ.get("BENCH_CONFIGS_VARIANTS"):   # development: layer-0 tile choice at this size
        from torchani_amd import _lib
        from torchani_amd.engine import PackedNetworks

        for nm, fl in (("big_tiles", _lib.MLP_FLAG_BIG_TILES), ("no_fused", _lib.MLP_FLAG_NO_FUSED)):
            PackedNetworks.default_flags = fl
            g2 = model.graphed(spd, xd)
            out["config2"]["ms_graph_" + nm] = timeit(lambda: g2(xd), reps=reps2) * 1e3
        PackedNetworks.default_flags = 0
    model3 = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
    for key, name in (("config3", "cfg3_1hz5_water_ani2x"), ("config3_1c17", "cfg3_1c17_ani2x")):
        with np.load(os.path.join(GOLD, name + ".npz")) as z:
            sp3, x3 = z["species"].astype(np.int64), z["coords"]
            cell = z["cell"] if "cell" in z.files else None
        s3, c3 = torch.from_numpy(sp3).to(dev), torch.from_numpy(x3).to(dev)
        cl = None if cell is None else torch.from_numpy(cell).to(dev)
        pbc = None if cell is None else (True, True, True)
        dt = timeit(lambda: model3.energies_and_forces(s3, c3, cl, pbc, check_overflow=False), reps=reps3)
        out[key] = {"workload": f"{name}: {sp3.size} atoms, {'periodic' if pbc else 'no PBC'}, cell mode",
                    "ms_per_step": dt * 1e3, "atom_steps_per_s": sp3.size / dt}
    return out


def measure_config5(steps: int = 10):
    """BASELINE config 5: one training step on a minibatch of 2560 conformers (tools/train_bench.py: the recipe of the
    reference's tools/training-aev-benchmark.py:71-170 -- MSE(E) / sqrt(n_atoms), Adam lr 1e-4), ANI-2x x 8 members and
    the reference's own single ANI-1x network (csrc/README.md:106-112 publishes 9.45 ms per 2560-batch on a V100 for it)."""
    import train_bench

    out = {}
    old = ["--optimizer", "torch", "--train-precision", "fp32"]   # rounds 1-4: exact-fp32 passes + torch.optim.Adam
    for tag, argv in (("ani2x_x8_eager", ["--kind", "ani2x", "--members", "8"]),
                      ("ani2x_x8_graph", ["--kind", "ani2x", "--members", "8", "--graph"]),
                      ("ani2x_x8_graph_fp32_torch_adam", ["--kind", "ani2x", "--members", "8", "--graph"] + old),
                      ("ani1x_x1_eager", ["--kind", "ani1x", "--members", "1"]),
                      ("ani1x_x1_graph", ["--kind", "ani1x", "--members", "1", "--graph"]),
                      ("ani1x_x1_graph_fp32_torch_adam", ["--kind", "ani1x", "--members", "1", "--graph"] + old)):
        out[tag] = train_bench.run(train_bench.parse(argv + ["--steps", str(steps), "--warmup", "3"]), quiet=True)
        torch.cuda.empty_cache()
    out["workload"] = ("2560 synthetic ANI-1x-like conformers (H C N O, 2-24 atoms, padded), energy loss MSE / sqrt(n_atoms), "
                       "Adam lr 1e-4; eager = AEV / networks / backward / optimizer timed separately, graph = the whole step "
                       "replayed as one HIP graph; default = the fast path of round 5 (fused split-fp16 forward + backward kernel, "
                       "bf16x3 weight gradients, torchani_amd.optim.Adam: one launch, gradients written in place), "
                       "*_fp32_torch_adam = the exact-fp32 layer-by-layer passes with torch.optim.Adam (rounds 1-4) in the same run")
    return out


if __name__ == "__main__":
    res = measure()
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        for k, v in res.items():
            print(k, json.dumps(v))
