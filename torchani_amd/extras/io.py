"""Reading and writing (ext)xyz files: ``read_xyz`` / ``write_xyz`` with the reference's signatures and conventions
(torchani/io.py:22-176): tensors of shape [C, A] (atomic numbers, padding -1) and [C, A, 3]; a ``Lattice="..."`` entry
of the comment line of the first conformation is the cell of all of them (pbc all true); atomic number 100 is the
placeholder some viewers need for padding atoms.  These are the tensors ``model((species, coordinates), cell, pbc)`` and
``model.energies_and_forces`` take."""
from __future__ import annotations

import shlex
import typing as tp
from pathlib import Path

import torch
from torch import Tensor

from ..utils import pad_atomic_properties

__all__ = ["read_xyz", "write_xyz", "TorchaniIOError", "PERIODIC_TABLE"]

# element symbols by atomic number (index 0: no element)
PERIODIC_TABLE: tp.Tuple[str, ...] = ("",) + tuple(
    "H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc "
    "Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi "
    "Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og".split())
_Z_OF = {s: z for z, s in enumerate(PERIODIC_TABLE) if s}


class TorchaniIOError(IOError):
    pass


def write_xyz(species: Tensor, coordinates: Tensor, dest, cell: tp.Optional[Tensor] = None, pad: bool = False,
              pad_coord_value: float = 0.0, pad_species_value: int = 100) -> None:
    """Write conformations [C, A] / [C, A, 3] as an extxyz file (io.py:22-78).  ``pad=False``: atoms with species -1 are
    left out; ``pad=True``: they are written as element ``pad_species_value`` at ``pad_coord_value`` so that every frame has
    the same number of atoms.  A cell goes into the comment line of every frame."""
    if species.dim() != 2:
        raise ValueError("Species should be a 2 dim tensor")
    if coordinates.shape != (species.shape[0], species.shape[1], 3):
        raise ValueError("Coordinates should have shape (molecules, atoms, 3)")
    if cell is not None and cell.shape != (3, 3):
        raise ValueError("Cell should be a tensor of shape (3, 3)")
    if pad and (species == pad_species_value).any():
        raise ValueError(f"Can't pad if there are elements with atomic number {pad_species_value}")
    props = "species:S:1:pos:R:3"
    header = f'Properties={props} pbc="F F F"\n'
    if cell is not None:
        elements = " ".join((f"{e:.10f}" if e != 0.0 else "0.0") for e in cell.detach().reshape(-1).tolist())
        header = f'Lattice="{elements}" Properties={props} pbc="T T T"\n'
    znums_all = species.detach().cpu().tolist()
    coords_all = coordinates.detach().cpu().tolist()
    with open(Path(dest).resolve(), mode="wt", encoding="utf-8") as f:
        for znums, coords in zip(znums_all, coords_all):
            atoms = [(z, xyz) if z != -1 else (pad_species_value, [pad_coord_value] * 3)
                     for z, xyz in zip(znums, coords) if pad or z != -1]
            f.write(f"{len(atoms)}\n")
            f.write(header)
            for z, (x, y, zc) in atoms:
                f.write(f"{PERIODIC_TABLE[z]} {x:.10f} {y:.10f} {zc:.10f}\n")


def read_xyz(path, dtype=None, device=None, detect_padding: bool = True, pad_species_value: int = 100,
             dividing_char: str = ">", return_comments: bool = False):
    """Read an xyz file with possibly many conformations (io.py:81-176) -> (species [C, A] int64 atomic numbers, coordinates
    [C, A, 3], cell [3, 3] or None, pbc [3] or None[, comments]).  Shorter conformations are padded with species -1 and
    zero coordinates; with ``detect_padding`` atoms of element ``pad_species_value`` become padding too.  The first column
    of an atom line is a symbol or an atomic number.  The cell is the ``Lattice`` of the first conformation; a later,
    different one is an error."""
    cell: tp.Optional[Tensor] = None
    properties: tp.List[tp.Dict[str, Tensor]] = []
    comments: tp.List[str] = []
    with open(Path(path).resolve(), mode="rt", encoding="utf-8") as f:
        lines = iter(f)
        n_conf = 0
        for head in lines:
            if dividing_char and head.strip() == dividing_char:
                continue
            num = int(head)
            comment = next(lines)
            if return_comments:
                comments.append(comment)
            if "lattice" in comment.lower():
                if cell is None and n_conf != 0:
                    raise TorchaniIOError("If cell is present it should be in the first conformation")
                for part in shlex.split(comment):
                    key, _, value = part.partition("=")
                    if key.lower() == "lattice":
                        this = torch.tensor([float(s) for s in value.split()], dtype=dtype, device=device).view(3, 3)
                        if cell is None:
                            cell = this
                        elif not (cell == this).all():
                            raise TorchaniIOError("Found two conformations with non-matching cells")
            znums, coords = [], []
            for _ in range(num):
                s, x, y, z = next(lines).split()[:4]
                zn = _Z_OF[s] if s in _Z_OF else int(s)
                if zn == pad_species_value and detect_padding:
                    zn, x, y, z = -1, "0.0", "0.0", "0.0"
                znums.append(zn)
                coords.append([float(x), float(y), float(z)])
            n_conf += 1
            properties.append({"coordinates": torch.tensor([coords], dtype=dtype, device=device),
                               "species": torch.tensor([znums], dtype=torch.long, device=device)})
    out = pad_atomic_properties(properties)
    pbc = torch.tensor([True, True, True], device=device) if cell is not None else None
    if return_comments:
        return out["species"], out["coordinates"], cell, pbc, comments
    return out["species"], out["coordinates"], cell, pbc
