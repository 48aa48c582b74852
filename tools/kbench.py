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
    ap.add_argument("--chunk", type=int, default=1 << 18, help="atoms per network chunk")
    args = ap.parse_args()
    from torchani_amd.models import ANI2x

    dev = torch.device("cuda:0")
    sp_np, x_np, cell_np = water_box(args.side)
    n = sp_np.shape[1]
    sp32 = torch.from_numpy(sp_np).to(dev).to(torch.int32).contiguous()
    coords = torch.from_numpy(x_np).to(dev)
    cell = torch.from_numpy(cell_np).to(dev)
    pbc = (True, True, True)
    model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
    eng = model.aev_computer.engine()
    packed = model.neural_networks._pack(dev)
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
    if "bwd" in st:
        out["bwd"] = time_stage(lambda: eng.backward(sp32, nbrs, gaev, gc), args.reps)
    if "mlp" in st:
        if args.mask != "on":
            out["mlp_dense"] = time_stage(lambda: packed.forward_backward(sp32, aev, atomic_e=ae, grad_aev=gaev),
                                          args.reps)
    if "mlp" in st and args.mask != "off":
        out["mlp"] = time_stage(lambda: packed.forward_backward(sp32, aev, atomic_e=ae, grad_aev=gaev,
                                                                slab_mask=mask, chunk=args.chunk), args.reps)
    bpa = 3584 + 20 * n_a + 8 + 448 + 8 * n_r
    line = f"atoms={n} n_r={n_r:.1f} n_a={n_a:.1f} | " + " ".join(f"{k}={v:.3f}ms" for k, v in out.items())
    if "fwd" in out:
        line += f" | fwd {bpa * n / out['fwd'] / 1e6:.0f} GB/s ({bpa * n / out['fwd'] / 1e6 / 8000:.1%} of 8 TB/s)"
    if "mlp" in out:
        line += f" | mlp {mlp_flops_per_atom(sp_np.reshape(-1)) * n / out['mlp'] / 1e9:.1f} TFLOP/s"
    print(line)


if __name__ == "__main__":
    main()
