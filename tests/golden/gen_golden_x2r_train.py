"""Golden fixtures for TRAINING the GELU / bias-free networks of the ANI-2xr family: the REFERENCE's autograd, first and
second order (the reference trains these networks through plain autograd, arch.py:992-1066, nn/_core.py:146-167).

    python tests/golden/gen_golden_x2r_train.py      (needs /root/reference; the outputs are committed)

For a base fixture the reference's simple_ani model of ANI-2xr (fp64, pyaev, seeded parameters; networks only: repulsion
and self energies do not depend on the parameters) gives
  * Loss_E = sum_{c,a} g[c,a] E_atomic[c,a]   (g of gen_golden_wgrads.upstream), back-propagated to every weight, and
  * Loss_F = sum_k t_k . F_k with F = -dE/dr under create_graph=True (t of gen_golden_fgrads.direction).
Stored: the losses and digests (gen_golden_wgrads.digest) of the gradient vectors flattened member -> species (the model's
order H C N O F S Cl) -> layer, weights only (the networks have no biases).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_d3 as g3  # noqa: E402,F401  (reference import + h5py stand-in)
import gen_golden as gg  # noqa: E402,F401

import torch  # noqa: E402
from gen_golden_fgrads import direction  # noqa: E402
from gen_golden_wgrads import digest, upstream  # noqa: E402
from torchani.arch import simple_ani  # noqa: E402

from torchani_amd.weights import arch_spec, random_state_dict  # noqa: E402

PREFIX = "potentials.nnp.neural_networks."
LAYERS = ("layers.0", "layers.1", "layers.2", "final_layer")


def flat_grads(model, symbols, n_members):
    named = dict(model.named_parameters())
    parts = []
    for m in range(n_members):
        for s in symbols:
            for lay in LAYERS:
                p = named[f"{PREFIX}members.{m}.atomics.{s}.{lay}.weight"]
                parts.append((p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().reshape(-1))
    return np.concatenate(parts)


def run(name, seed, n_members=8):
    with np.load(os.path.join(HERE, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = arch_spec("ani2xr")[0]
    old = [str(s) for s in g["symbols"]]
    remap = np.asarray([symbols.index(s) for s in old] + [-1])
    species = remap[g["species"]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = simple_ani(lot="wb97x-631gd", symbols=symbols, ensemble_size=n_members, dispersion=False, repulsion=True,
                           periodic_table_index=False)
    state = {k: torch.from_numpy(v) for k, v in random_state_dict("ani2xr", n_members, seed).items()}
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not [k for k in missing if "neural_networks" in k] and not unexpected
    model = model.double()
    nets = model.potentials["nnp"].neural_networks
    nets.requires_grad_(True)
    elem = torch.from_numpy(species.astype(np.int64))
    cell = torch.from_numpy(g["cell"]).double() if "cell" in g else None
    pbc = torch.from_numpy(g["pbc"]) if "pbc" in g else None
    C, A = elem.shape
    # first order: loss on the per-atom energies
    coords = torch.from_numpy(g["coords"]).double()
    aev = model.aev_computer(elem, coords, cell, pbc)
    atomic = nets(elem, aev, atomic=True)
    loss_e = (atomic * torch.as_tensor(upstream(C, A))).sum()
    loss_e.backward()
    ge = flat_grads(model, symbols, n_members)
    for p in nets.parameters():
        p.grad = None
    # second order: loss on the forces
    x = torch.from_numpy(g["coords"]).double().requires_grad_(True)
    t = torch.as_tensor(direction(C, A)) * (elem >= 0).unsqueeze(-1)
    e = nets(elem, model.aev_computer(elem, x, cell, pbc)).sum()
    (gx,) = torch.autograd.grad(e, x, create_graph=True)
    loss_f = -(gx * t).sum()
    loss_f.backward()
    gf = flat_grads(model, symbols, n_members)
    out = dict(base=np.asarray(name), seed=np.asarray(seed), n_members=np.asarray(n_members), symbols=np.asarray(symbols),
               species=species.astype(np.int64), n_params=np.asarray(ge.shape[0]),
               loss_e=np.asarray(loss_e.item()), loss_f=np.asarray(loss_f.item()))
    for tag, v in (("e", ge), ("f", gf)):
        sums, dots, heads = digest(v)
        out.update({f"{tag}_sums": sums, f"{tag}_dots": dots, f"{tag}_heads": heads, f"{tag}_abs_max": np.asarray(np.abs(v).max())})
    path = os.path.join(HERE, f"x2rtrain_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"x2rtrain_{name}: params={ge.shape[0]} loss_e={loss_e.item():+.9f} |ge|max={np.abs(ge).max():.4e} "
          f"loss_f={loss_f.item():+.9f} |gf|max={np.abs(gf).max():.4e} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    run("rand_batch_ani2x", 31)
    run("water_pbc_ani2x", 32)
