"""Per-stage device timing of the hot path on a periodic water box (development microbenchmark).

    python tools/kbench.py --side 56 [--reps 5] [--stages nbr,fwd,bwd,mlp]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import mlp_flops_per_atom, time_stage, water_box  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=56)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--stages", default="nbr,fwd,bwd,mlp")
    ap.add_argument("--mask", default="both", choices=["on", "off", "both"], help="slab masks in the mlp stage")
    ap.add_argument("--chunk", type=int, default=0, help="atoms per launch group of the network stage (0: the engine's rule)")
    ap.add_argument("--mlp-flags", type=int, default=0, help="anihip_mlp_desc.flags (ANIHIP_MLP_FLAG_*) of the network stage")
    ap.add_argument("--compact", action="store_true", help="species numbered present-ones-first (models.ANI.compact_species)")
    ap.add_argument("--order", default="lattice", help="atom order: lattice (as generated), shuffle, layers (quarter-cutoff "
                    "layers along x, then cutoff cells), brick:<B> (bricks of B x B x B cutoff cells, cells inside in z-fastest order)")
    args = ap.parse_args()
    from torchani_amd.models import ANI2x

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    sp_np, x_np, cell_np = water_box(args.side)
    n = sp_np.shape[1]
    if args.order != "lattice":
        import numpy as np

        x = x_np.reshape(-1, 3)
        box = float(cell_np[0, 0])
        if args.order == "shuffle":
            perm = np.random.RandomState(1).permutation(n)
        elif args.order == "layers":
            k0 = np.floor(x[:, 0] / (5.1 / 4)).astype(np.int64)
            k1 = np.floor(x[:, 1] / 5.1).astype(np.int64)
            k2 = np.floor(x[:, 2] / 5.1).astype(np.int64)
            perm = np.argsort((k0 * 1024 + k1) * 1024 + k2, kind="stable")
        else:
            B = int(args.order.split(":")[1])
            c = np.floor(x / (box / np.floor(box / 5.1))).astype(np.int64)
            b, w = c // B, c % B
            key = ((((b[:, 0] * 256 + b[:, 1]) * 256 + b[:, 2]) * 64 + w[:, 0]) * 64 + w[:, 1]) * 64 + w[:, 2]
            perm = np.argsort(key, kind="stable")
        sp_np, x_np = np.ascontiguousarray(sp_np[:, perm]), np.ascontiguousarray(x_np[:, perm])
    sp32 = torch.from_numpy(sp_np).to(dev).to(torch.int32).contiguous()
    coords = torch.from_numpy(x_np).to(dev)
    cell = torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
    eng = model.aev_computer.engine()
    order = None
    if args.compact:
        model.compact_species = True
        sp32, order = model._engine_species(sp32)
    packed = model.neural_networks._pack(dev, order)
    if args.mlp_flags:
        packed.flags = args.mlp_flags
    nbrs = eng.neighbors(sp32, coords, cell, pbc, mode="cell")
    mask = torch.zeros(n, dtype=torch.int32, device=dev)
    aev = eng.forward(sp32, nbrs, slab_mask=mask)
    ae = torch.zeros(n, dtype=torch.float32, device=dev)
    gaev = torch.randn_like(aev) * 1e-3
    gc = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    meta = nbrs.meta[:, 1].to(torch.int64) & 0xFFFFFFFF
    n_a = float((meta & 0xFFFF).double().mean())
    n_r = n_a + float((meta >> 16).double().mean())
    st = args.stages.split(",")
    out = {}
    if "nbr" in st:
        out["nbr"] = time_stage(lambda: eng.neighbors(sp32, coords, cell, pbc, mode="cell"), args.reps)
    if "fwd" in st:
        out["fwd"] = time_stage(lambda: eng.forward(sp32, nbrs, out=aev, slab_mask=mask), args.reps)
    if "fwdu" in st:   # rows updated in place in the engine's kept buffers (the product path of energies_and_forces)
        eng.forward_update(sp32, nbrs, shard_rows=False)
        out["fwd_update"] = time_stage(lambda: eng.forward_update(sp32, nbrs, shard_rows=False), args.reps)
    if "bwd" in st:
        out["bwd"] = time_stage(lambda: eng.backward(sp32, nbrs, gaev, gc, slab_mask=mask), args.reps)   # (as in the product path)
    if "mlp" in st:
        if args.mask != "on":
            out["mlp_dense"] = time_stage(lambda: packed.forward_backward(sp32, aev, atomic_e=ae, grad_aev=gaev),
                                          args.reps)
    if "mlp" in st and args.mask != "off":
        out["mlp"] = time_stage(lambda: packed.forward_backward(sp32, aev, atomic_e=ae, grad_aev=gaev,
                                                                slab_mask=mask, chunk=args.chunk or None), args.reps)
    bpa = 3584 + 20 * n_a + 8 + 448 + 8 * n_r
    line = f"atoms={n} n_r={n_r:.1f} n_a={n_a:.1f} | " + " ".join(f"{k}={v:.3f}ms" for k, v in out.items())
    if "fwd" in out:
        line += f" | fwd {bpa * n / out['fwd'] / 1e6:.0f} GB/s ({bpa * n / out['fwd'] / 1e6 / 8000:.1%} of 8 TB/s)"
    if "fwd_update" in out:
        line += f" | fwd_update {bpa * n / out['fwd_update'] / 1e6:.0f} GB/s ({bpa * n / out['fwd_update'] / 1e6 / 8000:.1%} of 8 TB/s on the same algorithmic bytes)"
    if "mlp" in out:
        line += f" | mlp {mlp_flops_per_atom(sp_np.reshape(-1)) * n / out['mlp'] / 1e9:.1f} TFLOP/s"
        # (checksums: library variants that only reschedule instructions must print the same digits)
        line += f" | sum(e)={float(ae.double().sum()):.9f} sum|dE/dAEV|={float(gaev.double().abs().sum()):.7f}"
    print(line)


if __name__ == "__main__":
    main()
