"""Golden fixtures of the DFT-D3(BJ) two-body dispersion term, from the reference's TwoBodyDispersionD3 in float64
(torchani/potentials/dftd3.py; envelope / halves of potentials/core.py) on the coordinates of existing golden cases,
and the element data the engine's own TwoBodyDispersionD3 ships with:

    python tests/golden/gen_golden_d3.py      -> tests/golden/d3_<case>.npz, torchani_amd/data/d3_refs.npz

The reference reads its C6 reference table (Grimme's D3 parameters) from resources/c6.h5 through h5py, which is not
installed here: ``read_c6_h5`` below is a reader for exactly that file (HDF5 superblock 0, three contiguous little-endian
float32 datasets all/{constants, coordnums_a, coordnums_b} of shape [95, 95, 5, 5]; the data-layout messages -- version
3, class 1 = contiguous, address, size -- are located by their size field) and stands in for ``h5py.File`` while the
reference module is imported.  d3_refs.npz holds the slice Z <= 18 of the table plus the covalent radii and
sqrt(empirical charge) values of resources/atomic_constants.json and resources/functional_d3bj_constants.json: data
(published D3 parameters), not code.

Each fixture: per-atom energies (atomic=True), molecular energies and forces with cutoff 8.0 A and the smooth
envelope, functional wB97X (the recipe of arch.py:1177-1181 for a wB97X model), and for water_pbc also b973c.
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the reference import, with a dummy h5py module)

RES = "/root/reference/torchani/resources"
SHAPE = (95, 95, 5, 5)
NBYTES = 4 * int(np.prod(SHAPE))


def read_c6_h5(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89HDF\r\n\x1a\n" and b[8] == 0, "not the HDF5 file this reader was written for"
    addrs = []
    pat = struct.pack("<Q", NBYTES)
    i = b.find(pat)
    while i >= 0:
        if b[i - 10] == 3 and b[i - 9] == 1:   # layout message: version 3, contiguous
            addrs.append(struct.unpack("<Q", b[i - 8:i])[0])
        i = b.find(pat, i + 1)
    assert len(addrs) == 3, addrs
    # the object headers were written in the order the names appear in the group's heap
    names = sorted(("constants", "coordnums_a", "coordnums_b"), key=lambda n: b.find(n.encode()))
    assert names == ["constants", "coordnums_a", "coordnums_b"]
    out = {n: np.frombuffer(b, dtype="<f4", count=NBYTES // 4, offset=a).reshape(SHAPE).copy()
           for n, a in zip(names, sorted(addrs))}
    c6 = out["constants"]
    assert abs(c6[1, 1, 0, 0] - 3.0267) < 1e-4 and abs(c6[6, 6, 0, 0] - 49.113) < 0.5, "H-H / C-C C6 references"
    return out


class _H5File:
    """Stand-in for h5py.File(path, 'r') over read_c6_h5."""

    def __init__(self, path, mode="r"):
        self._d = {"all/" + k: v for k, v in read_c6_h5(path).items()}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __getitem__(self, k):
        return self._d[k]


sys.modules["h5py"].File = _H5File

import torch  # noqa: E402
from torchani.neighbors import all_pairs  # noqa: E402
from torchani.potentials import TwoBodyDispersionD3  # noqa: E402


def write_refs():
    t = read_c6_h5(os.path.join(RES, "c6.h5"))
    Z = 19
    ac = json.load(open(os.path.join(RES, "atomic_constants.json")))
    sym = [s for s in ac if s][:Z - 1]
    cov = np.zeros(Z, np.float64)
    sq = np.zeros(Z, np.float64)
    for z, s in enumerate(sym, start=1):
        cov[z] = ac[s]["covalent_radius"] if ac[s].get("covalent_radius") is not None else np.nan
        sq[z] = ac[s]["sqrt_empirical_charge"] if ac[s].get("sqrt_empirical_charge") is not None else np.nan
    fn = json.load(open(os.path.join(RES, "functional_d3bj_constants.json")))
    names = sorted(fn)
    path = os.path.join(gg.ROOT, "torchani_amd", "data", "d3_refs.npz")
    np.savez_compressed(path, symbols=np.asarray(sym), c6=t["constants"][:Z, :Z], cn_a=t["coordnums_a"][:Z, :Z],
                        cn_b=t["coordnums_b"][:Z, :Z], covalent_radius=cov, sqrt_empirical_charge=sq,
                        functionals=np.asarray(names),
                        functional_s6_s8_a1_a2=np.asarray([[fn[k]["s6"], fn[k]["s8"], fn[k]["a1"], fn[k]["a2"]] for k in names]))
    print(path, os.path.getsize(path), "bytes")


def run(name, functional, tag, cutoff=8.0, cutoff_fn="smooth"):
    with np.load(os.path.join(HERE, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    symbols = [str(s) for s in g["symbols"]]
    pot = TwoBodyDispersionD3.from_functional(symbols=symbols, functional=functional, cutoff=cutoff,
                                              cutoff_fn=cutoff_fn).double()
    elem = torch.from_numpy(g["species"].astype(np.int64))
    coords = torch.from_numpy(g["coords"]).double().requires_grad_(True)
    cell = torch.from_numpy(g["cell"]).double() if "cell" in g else None
    pbc = torch.from_numpy(g["pbc"]) if "pbc" in g else None
    neighbors = all_pairs(cutoff, elem, coords, cell, pbc)
    atomic = pot.compute_from_neighbors(elem, coords, neighbors, atomic=True).energies
    e = atomic.sum(dim=1)
    (grad,) = torch.autograd.grad(e.sum(), coords)
    out = dict(cutoff=np.asarray(cutoff), cutoff_fn=np.asarray(cutoff_fn), functional=np.asarray(functional),
               atomic_energies=atomic.detach().numpy(), energies=e.detach().numpy(), forces=(-grad).numpy())
    path = os.path.join(HERE, f"d3_{tag}{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: pairs={neighbors.indices.shape[1]} E[0]={e[0].item():+.9f} |F|max={grad.abs().max().item():.6f}")


if __name__ == "__main__":
    write_refs()
    for nm in ("rand_batch_ani2x", "water_pbc_ani2x", "small_ani2x", "triclinic_pbc_ani2x"):
        run(nm, "wb97x", "")
    run("water_pbc_ani2x", "b973c", "b973c_")
