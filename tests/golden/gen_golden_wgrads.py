"""Golden weight-gradient fixtures: the REFERENCE's autograd applied to a loss on the per-atom energies.

    python tests/golden/gen_golden_wgrads.py      (needs /root/reference; the outputs are committed)

For a base fixture (species / coords / cell / seed of tests/golden/<base>.npz) the reference model (fp64, pyaev,
the seeded parameters of gen_golden.py) is evaluated with ``atomic=True`` and

    Loss = sum_{c,a} g[c,a] * E_atomic[c,a],      g[c,a] = 0.5 + frac(0.37 * (c*A + a))

is back-propagated to every weight and bias of the 8-member ensemble (what the training loop of
tools/training-aev-benchmark.py:120-135 does with an MSE loss).  The gradient vector in the oracle's packed
layout (oracle.pack_networks) has up to 13.7 M entries, so the fixture keeps a digest: per block of 4096 entries
the plain sum and the dot product with cos(0.37 k) (k = global index), plus the first 64 entries of every block
whose index is a multiple of 97.  tests/test_oracle_golden.py recomputes the digest from the oracle's gradients.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402

from oracle import oracle as orc  # noqa: E402

BLOCK = 4096


def upstream(C, A):
    k = np.arange(C * A, dtype=np.float64)
    return (0.5 + np.modf(0.37 * k)[0]).reshape(C, A)


def digest(flat):
    n = flat.shape[0]
    nb = (n + BLOCK - 1) // BLOCK
    pad = np.zeros(nb * BLOCK, dtype=np.float64)
    pad[:n] = flat
    pat = np.cos(0.37 * np.arange(nb * BLOCK, dtype=np.float64))
    sums = pad.reshape(nb, BLOCK).sum(axis=1)
    dots = (pad * pat).reshape(nb, BLOCK).sum(axis=1)
    heads = pad.reshape(nb, BLOCK)[::97, :64].copy()
    return sums, dots, heads


def run_case(base):
    torch = gg.torch
    with np.load(os.path.join(gg.HERE, base + ".npz")) as z:
        b = {k: z[k] for k in z.files}
    kind, seed = str(b["kind"]), int(b["seed"])
    gg.CUTOFF_FN = str(b["cutoff_fn"]) if "cutoff_fn" in b else "cosine"
    model = gg.build_reference(kind, seed)
    nets = model.potentials["nnp"].neural_networks if hasattr(model, "potentials") else model.neural_networks
    nets.requires_grad_(True)
    elem = torch.as_tensor(b["species"].astype(np.int64))
    coords = torch.as_tensor(b["coords"]).double()
    cell = torch.as_tensor(b["cell"]).double() if "cell" in b else None
    pbc = torch.as_tensor(b["pbc"]) if "pbc" in b else None
    aev = model.aev_computer(elem, coords, cell, pbc)
    atomic = nets(elem, aev, atomic=True)
    C, A = elem.shape
    g = upstream(C, A)
    loss = (atomic * torch.as_tensor(g)).sum()
    loss.backward()
    prefix = "potentials.nnp.neural_networks."
    grads = {}
    for name, p in model.named_parameters():
        if name.startswith(prefix):
            grads[name] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    symbols = [str(s) for s in b["symbols"]]
    dims, flat = orc.pack_networks(grads, symbols, 8)
    sums, dots, heads = digest(flat)
    path = os.path.join(gg.HERE, "wgrads_" + base + ".npz")
    np.savez_compressed(path, base=np.asarray(base), loss=np.asarray(loss.item()), n_params=np.asarray(flat.shape[0]),
                        block_sums=sums, block_dots=dots, block_heads=heads,
                        grad_abs_max=np.asarray(np.abs(flat).max()), grad_l2=np.asarray(np.linalg.norm(flat)))
    print(f"wgrads_{base}: params={flat.shape[0]} loss={loss.item():+.9f} |g|max={np.abs(flat).max():.4e} "
          f"|g|2={np.linalg.norm(flat):.6e} -> {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    gg.torch.set_num_threads(8)
    for base in ("ch4_ani1x", "rand_batch_ani2x", "water_pbc_smooth_ani2x", "dense90_ani2x"):
        run_case(base)


if __name__ == "__main__":
    main()
