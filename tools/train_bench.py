"""Training-step timing for BASELINE.json config 5 (the recipe of the reference's tools/training-aev-benchmark.py:
minibatch of 2560 conformers, MSE(E) / sqrt(n_atoms) loss, Adam lr 1e-4; stages timed separately like
csrc/README.md:108-112):

    python tools/train_bench.py [--kind ani1x|ani2x] [--members M] [--batch 2560] [--atoms 24]

Synthetic ANI-1x-like conformers (H C N O, 2..atoms real atoms, padded with -1), seeded random weights.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def conformers(n_mol, n_at, seed=5):
    rs = np.random.RandomState(seed)
    sp = np.full((n_mol, n_at), -1, dtype=np.int64)
    x = np.zeros((n_mol, n_at, 3), dtype=np.float32)
    # a jittered lattice of spacing 1.1 A gives organic-molecule-like neighbor counts without rejection sampling
    side = int(np.ceil(n_at ** (1 / 3)))
    grid = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    for m in range(n_mol):
        k = rs.randint(2, n_at + 1)
        pick = rs.permutation(len(grid))[:k]
        x[m, :k] = 1.1 * grid[pick] + rs.uniform(-0.15, 0.15, (k, 3)).astype(np.float32)
        sp[m, :k] = rs.choice([0, 1, 2, 3], size=k, p=[0.5, 0.3, 0.1, 0.1])
    return sp, x


def step_flops(sp, kind, members):
    """MFMA work of one energy-training step on the fast path, from the batch's species: fp32-equivalent flops (2 per
    multiply-add of the mathematical GEMMs: layer 0 over the AEV slabs of species (pairs) that occur in the batch, hidden
    layers forward and backward, weight gradients over the same slabs) and the bf16 / fp16 flops actually issued (the
    fused kernel's three-product split, the weight-gradient kernel's six products)."""
    from torchani_amd.weights import arch_spec

    symbols, consts, hidden = arch_spec(kind)
    S = len(symbols)
    present = [int((sp == s).sum()) for s in range(S)]
    n_present = sum(1 for c in present if c > 0)
    rad_slabs = len({s // 2 for s in range(S) if present[s] > 0})
    slabs = rad_slabs + n_present * (n_present + 1) // 2 if consts.out_dim == 16 * S + 32 * (S * (S + 1) // 2) else -(-consts.out_dim // 32)
    fb = wg = 0.0
    for s, sym in enumerate(symbols):
        h1, h2, h3 = hidden[sym]
        r = lambda v: -(-v // 32) * 32   # (the kernels work on widths padded to 32)
        h1, h2, h3 = r(h1), r(h2), r(h3)
        fb += present[s] * members * 2.0 * (32 * slabs * h1 + 2 * h1 * h2 + 2 * h2 * h3 + h3)
        # (the layer-0 weight gradients run over the same flagged slabs: the columns of absent species are zero)
        wg += present[s] * members * 2.0 * (32 * slabs * h1 + h1 * h2 + h2 * h3 + h3)
    return {"fp32_equivalent": fb + wg, "issued": 3.0 * fb + 6.0 * wg, "flagged_slabs": slabs,
            "forward_backward_fp32_equivalent": fb, "weight_grads_fp32_equivalent": wg}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="ani1x")
    ap.add_argument("--members", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2560)
    ap.add_argument("--atoms", type=int, default=24)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--forces", action="store_true", help="energy + force loss (second-order backward)")
    ap.add_argument("--graph", action="store_true", help="capture the whole step (forward, backward, Adam) in a HIP graph")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="fused: torchani_amd.optim.Adam (one launch over flat buffers, gradients written in place); torch: torch.optim.Adam")
    ap.add_argument("--train-precision", default="f16x3", choices=["f16x3", "fp32"],
                    help="f16x3: fused forward+backward kernel and bf16x3 weight gradients; fp32: the exact-fp32 layer-by-layer passes")
    return ap.parse_args(argv)


def run(args, quiet=False):
    """One configuration; returns the numbers that are printed (bench.py's secondary.config5 calls this)."""
    from torchani_amd.models import ANI1x, ANI2x

    dev = torch.device("cuda:0")
    ctor = ANI2x if args.kind == "ani2x" else ANI1x
    model = ctor(seed=0, n_members=args.members, device=dev, periodic_table_index=False, neighborlist="batch")
    nets = model.neural_networks
    nets.requires_grad_(True)
    nets.train_precision = getattr(args, "train_precision", "f16x3")
    if getattr(args, "optimizer", "fused") == "fused":
        from torchani_amd.optim import Adam

        opt = Adam(nets.parameters(), lr=1e-4)
    else:
        opt = torch.optim.Adam(nets.parameters(), lr=1e-4, capturable=args.graph)
    sp, x = conformers(args.batch, args.atoms)
    spd, xd = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
    n_at = (spd >= 0).sum(dim=1).float()
    target = torch.from_numpy(np.random.RandomState(1).normal(0, 0.1, args.batch).astype(np.float32)).to(dev)
    true_f = torch.from_numpy(np.random.RandomState(2).normal(0, 0.05, x.shape).astype(np.float32)).to(dev)
    true_f = true_f * (spd >= 0).unsqueeze(-1)
    acc = dict(aev=0.0, nn=0.0, force=0.0, backward=0.0, optimizer=0.0)

    def step(timed):
        def mark():
            torch.cuda.synchronize()
            return time.perf_counter()
        xin = xd.clone().requires_grad_(True) if args.forces else xd
        t0 = mark()
        aev = model.aev_computer(spd, xin)
        t1 = mark()
        e = nets(spd, aev)
        t2 = mark()
        loss = (torch.nn.functional.mse_loss(e, target, reduction="none") / n_at.sqrt()).mean()
        tf = t2
        if args.forces:   # tools/training-aev-benchmark.py:136-150: forces with create_graph, force coefficient 0.1
            forces = -torch.autograd.grad(e.sum(), xin, create_graph=True, retain_graph=True)[0]
            floss = (torch.nn.functional.mse_loss(true_f, forces, reduction="none").sum(dim=(1, 2)) / (3.0 * n_at)).mean()
            loss = loss + 0.1 * floss
            tf = mark()
        opt.zero_grad()
        loss.backward()
        t3 = mark()
        opt.step()
        t4 = mark()
        if timed:
            for k, v in zip(acc, (t1 - t0, t2 - t1, tf - t2, t3 - tf, t4 - t3)):
                acc[k] += v
        return float(loss.detach())

    if args.graph:
        # whole-step capture (PyTorch's "whole network" recipe): every launch of the C ABI goes to torch's current
        # stream and nothing in the step synchronises with the host, so forward, backward, the parameter refresh and
        # Adam replay as ONE graph launch
        def whole():
            aev = model.aev_computer(spd, xd)
            e = nets(spd, aev)
            loss = (torch.nn.functional.mse_loss(e, target, reduction="none") / n_at.sqrt()).mean()
            loss.backward()
            opt.step()
            return loss
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                whole()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = whole()
        for _ in range(args.warmup):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = None
        for i in range(args.steps):
            graph.replay()
            if i == 0:
                first = float(static_loss.detach())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        n_real = int((sp >= 0).sum())
        if not quiet:
            print(f"config 5, whole step replayed as a HIP graph: {args.kind} x{args.members}, batch {args.batch} conformers "
                  f"({n_real} atoms): {dt * 1e3:.2f} ms/step = {args.batch / dt:.0f} conformers/s; "
                  f"loss {first:.5f} -> {float(static_loss.detach()):.5f}")
        out = {"ms_per_step": dt * 1e3, "conformers_per_s": args.batch / dt, "atoms": n_real,
               "loss_first": first, "loss_last": float(static_loss.detach())}
        if nets.train_precision == "f16x3" and not args.forces:
            fl = step_flops(sp, args.kind, args.members)
            out["roofline"] = {"bound": "mfma", "kernels": "k_mlp_fused<2,1,CELU,TRAIN> (f16x3) + k_wgrad_b3 x 3 (bf16x3)",
                               "achieved": fl["issued"] / dt / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                               "frac": fl["issued"] / dt / 1e12 / 2500.0,
                               "fp32_equivalent_tflops": fl["fp32_equivalent"] / dt / 1e12, "fp32_mfma_peak": 157.3,
                               "flagged_slabs": fl["flagged_slabs"],
                               "note": "whole graphed step (AEV, networks forward + backward, weight gradients, Adam, parameter "
                                       "refresh) in the denominator; issued = 3 fp16 MFMA flops per fp32 flop of the fused "
                                       "kernel, 6 bf16 per fp32 flop of the weight gradients"}
        return out
    for _ in range(args.warmup):
        step(False)
    losses = [step(True) for _ in range(args.steps)]
    total = sum(acc.values()) / args.steps
    n_real = int((sp >= 0).sum())
    if not quiet:
        print(f"config 5{' (energy+force loss)' if args.forces else ''}: {args.kind} x{args.members}, batch {args.batch} conformers ({n_real} atoms): "
              f"{total * 1e3:.2f} ms/step = {args.batch / total:.0f} conformers/s")
        print("  " + "  ".join(f"{k} {v / args.steps * 1e3:.2f} ms" for k, v in acc.items()))
        print(f"  loss {losses[0]:.5f} -> {losses[-1]:.5f}")
    out = {"ms_per_step": total * 1e3, "conformers_per_s": args.batch / total, "atoms": n_real,
           "loss_first": losses[0], "loss_last": losses[-1]}
    out.update({f"ms_{k}": v / args.steps * 1e3 for k, v in acc.items()})
    return out


if __name__ == "__main__":
    run(parse())
