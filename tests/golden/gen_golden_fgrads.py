"""Golden fixtures for training on forces: the REFERENCE's autograd, second order.

    python tests/golden/gen_golden_fgrads.py      (needs /root/reference; the outputs are committed)

For a base fixture the reference model (fp64, pyaev, seeded parameters) gives forces F = -dE/dr with
``create_graph=True`` (the force branch of tools/training-aev-benchmark.py:136-150) and

    Loss = sum_k t_k . F_k,        t[c,a,:] = (frac(0.37 q), frac(0.61 q) - 0.5, 0.25 - frac(0.13 q)),  q = 3 (c A + a)

is back-propagated to every weight and bias.  Stored: the loss, J t = d aev / d r . t at the rows ``aev_rows`` of the
base fixture (what the reference's cuaev double backward returns, csrc/aev.cu:1986-2015 -- here from
torch.autograd.functional.jvp of the reference AEVComputer) and the digest of the parameter gradients
(gen_golden_wgrads.digest).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402
from gen_golden_wgrads import digest  # noqa: E402

from oracle import oracle as orc  # noqa: E402


def direction(C, A):
    q = 3.0 * np.arange(C * A, dtype=np.float64)
    t = np.stack([np.modf(0.37 * q)[0], np.modf(0.61 * q)[0] - 0.5, 0.25 - np.modf(0.13 * q)[0]], axis=-1)
    return t.reshape(C, A, 3)


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
    coords = torch.as_tensor(b["coords"]).double().requires_grad_(True)
    cell = torch.as_tensor(b["cell"]).double() if "cell" in b else None
    pbc = torch.as_tensor(b["pbc"]) if "pbc" in b else None
    C, A = elem.shape
    t = torch.as_tensor(direction(C, A))
    t = t * (elem >= 0).unsqueeze(-1)          # padding atoms carry no direction
    aev = model.aev_computer(elem, coords, cell, pbc)
    e = nets(elem, aev).sum()
    (g,) = torch.autograd.grad(e, coords, create_graph=True)
    loss = -(g * t).sum()
    loss.backward()
    prefix = "potentials.nnp.neural_networks."
    grads = {n: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
             for n, p in model.named_parameters() if n.startswith(prefix)}
    symbols = [str(s) for s in b["symbols"]]
    _, flat = orc.pack_networks(grads, symbols, 8)
    sums, dots, heads = digest(flat)
    _, jt = torch.autograd.functional.jvp(lambda x: model.aev_computer(elem, x, cell, pbc), coords.detach(), t)
    rows = b["aev_rows"]
    path = os.path.join(gg.HERE, "fgrads_" + base + ".npz")
    np.savez_compressed(path, base=np.asarray(base), loss=np.asarray(loss.item()), n_params=np.asarray(flat.shape[0]),
                        block_sums=sums, block_dots=dots, block_heads=heads,
                        grad_abs_max=np.asarray(np.abs(flat).max()), grad_l2=np.asarray(np.linalg.norm(flat)),
                        aev_jvp=jt.detach().numpy().reshape(C * A, -1)[rows])
    print(f"fgrads_{base}: loss={loss.item():+.9f} |g|max={np.abs(flat).max():.4e} |Jt|max={jt.abs().max():.3f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    gg.torch.set_num_threads(8)
    for base in ("ch4_ani1x", "rand_batch_ani2x", "water_pbc_ani2x", "water_pbc_smooth_ani2x"):
        run_case(base)


if __name__ == "__main__":
    main()
