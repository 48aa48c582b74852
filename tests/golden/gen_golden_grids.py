"""Golden fixtures for symmetry-function grids OTHER than the published 16 / 8x4 / 4x8 ones, from the REFERENCE's
``AEVComputer.from_constants`` (aev/_computer.py:602-666, pyaev strategy, float64): AEV rows and the vector-Jacobian
product with seeded cotangents.  Run only where /root/reference exists; the *.npz are committed.

    python tests/golden/gen_golden_grids.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (stubs h5py / zarr and puts /root/reference on the path)

import torch  # noqa: E402
from torchani.aev import AEVComputer  # noqa: E402

GRIDS = {
    # name: (num_species, Rcr, Rca, EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ, cutoff_fn)
    "r8_a4z4": (4, 5.2, 3.5, 16.0, np.linspace(0.9, 4.66, 8), 8.0, 32.0, np.linspace(0.9, 2.85, 4),
                (np.arange(4) + 0.5) * np.pi / 4, "cosine"),
    "r24_a10z8": (3, 4.6, 3.1, 30.0, np.linspace(0.8, 4.4, 24), 12.5, 14.1, np.linspace(0.8, 2.9, 10),
                  (np.arange(8) + 0.5) * np.pi / 8, "smooth"),
    "r5_a3z5": (7, 5.1, 3.5, 19.7, np.linspace(1.0, 4.6, 5), 12.5, 9.0, np.linspace(1.0, 3.0, 3),
                (np.arange(5) + 0.5) * np.pi / 5, "cosine"),
}


def reference_aev(grid, species, coords32, cell, pbc, seed):
    S, Rcr, Rca, EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ, cut = GRIDS[grid]
    f32 = lambda v: [float(np.float32(x)) for x in v]   # noqa: E731  (fp32-rounded constants, evaluated in fp64)
    aevc = AEVComputer.from_constants(float(np.float32(Rcr)), float(np.float32(Rca)), float(np.float32(EtaR)), f32(ShfR),
                                      float(np.float32(EtaA)), float(np.float32(Zeta)), f32(ShfA), f32(ShfZ), S,
                                      cutoff_fn=cut, neighborlist="all_pairs").double()
    x = torch.as_tensor(coords32).double().requires_grad_(True)
    cell_t = None if cell is None else torch.as_tensor(cell).double()
    pbc_t = None if pbc is None else torch.as_tensor(pbc, dtype=torch.bool)
    aev = aevc(torch.as_tensor(species, dtype=torch.long), x, cell_t, pbc_t)
    w = np.random.RandomState(seed).uniform(-1.0, 1.0, tuple(aev.shape)).astype(np.float32).astype(np.float64)   # (fp32-representable)
    (vjp,) = torch.autograd.grad((aev * torch.as_tensor(w)).sum(), x)
    # forward-mode derivative J t along the direction of gen_golden_fgrads.direction (what the reference's cuaev double
    # backward returns, csrc/aev.cu:1986-2015): pins the JVP of the general kernels
    from gen_golden_fgrads import direction
    C, A = np.asarray(species).shape
    t = torch.as_tensor(direction(C, A)) * (torch.as_tensor(species, dtype=torch.long) >= 0).unsqueeze(-1)
    _, jt = torch.autograd.functional.jvp(lambda y: aevc(torch.as_tensor(species, dtype=torch.long), y, cell_t, pbc_t),
                                          x.detach(), t)
    return aev.detach().numpy(), w, vjp.numpy(), jt.detach().numpy()


def save(name, grid, species, coords32, cell, pbc, seed):
    aev, w, vjp, jt = reference_aev(grid, species, coords32, cell, pbc, seed)
    S, Rcr, Rca, EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ, cut = GRIDS[grid]
    out = dict(num_species=np.asarray(S), Rcr=np.float32(Rcr), Rca=np.float32(Rca), EtaR=np.float32(EtaR),
               ShfR=np.asarray(ShfR, dtype=np.float32), EtaA=np.float32(EtaA), Zeta=np.float32(Zeta),
               ShfA=np.asarray(ShfA, dtype=np.float32), ShfZ=np.asarray(ShfZ, dtype=np.float32), cutoff_fn=np.asarray(cut),
               species=np.asarray(species, dtype=np.int32), coords=coords32, aev=aev, cotangent=w.astype(np.float32),
               aev_vjp=vjp, aev_jvp=jt)
    if cell is not None:
        out["cell"] = np.asarray(cell, dtype=np.float32)
        out["pbc"] = np.asarray(pbc, dtype=bool)
    path = os.path.join(HERE, f"grid_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"grid_{name}: L={aev.shape[-1]} |aev|max={np.abs(aev).max():.3f} |vjp|max={np.abs(vjp).max():.3f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def molecules(rs, n_mol, n_atoms, n_species, box, min_dist, pad_prob):
    Z, X = gg.random_molecules(rs, n_mol, n_atoms, n_species, box, min_dist, pad_prob)
    lut = {z: k for k, z in enumerate([1, 6, 7, 8, 16, 9, 17])}
    sp = np.vectorize(lambda z: lut.get(int(z), -1))(Z).astype(np.int32)
    return sp, X


def main():
    torch.set_num_threads(8)
    rs = np.random.RandomState(21)
    # padded batch of molecules, 4 species
    sp, X = molecules(rs, 5, 12, 4, 4.0, 0.8, 0.3)
    save("r8_a4z4_batch", "r8_a4z4", sp, X, None, None, 101)
    # periodic box, 3 species, smooth cutoff, 80 angular terms per block
    cell = np.diag([9.6, 10.4, 11.2]).astype(np.float32)
    sp, X = molecules(rs, 1, 60, 3, 9.6, 0.9, 0.0)
    X[0] *= np.array([1.0, 10.4 / 9.6, 11.2 / 9.6], dtype=np.float32)
    save("r24_a10z8_pbc", "r24_a10z8", sp, X, cell, [True, True, True], 102)
    # all 7 species, odd sizes, dense cluster (many pairs per block), no pbc
    sp, X = molecules(rs, 2, 40, 7, 3.4, 0.7, 0.1)
    save("r5_a3z5_dense", "r5_a3z5", sp, X, None, None, 103)


if __name__ == "__main__":
    main()
