"""ctypes front-end of the CPU oracle (oracle/ani_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of ani_oracle.c.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; torchani_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import typing as tp

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_SHIFTS = 64


class Params(C.Structure):
    _fields_ = [
        ("S", C.c_int),
        ("nR", C.c_int),
        ("nA", C.c_int),
        ("nZ", C.c_int),
        ("Rcr", C.c_double),
        ("Rca", C.c_double),
        ("EtaR", C.c_double),
        ("EtaA", C.c_double),
        ("Zeta", C.c_double),
        ("ShfR", C.c_double * MAX_SHIFTS),
        ("ShfA", C.c_double * MAX_SHIFTS),
        ("ShfZ", C.c_double * MAX_SHIFTS),
        ("cutoff_kind", C.c_int),
    ]


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (oracle/Makefile)."""
    libs = [os.path.join(_HERE, f"libani_oracle_{k}.so") for k in ("f64", "f32")]
    src = os.path.join(_HERE, "ani_oracle.c")
    stale = force or any(
        (not os.path.exists(p)) or os.path.getmtime(p) < os.path.getmtime(src) for p in libs
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def f32_consts(x: tp.Sequence[float]) -> tp.List[float]:
    """fp32-rounded constants promoted to double."""
    return [float(np.float32(v)) for v in x]


def linspace(start: float, stop: float, steps: int) -> tp.List[float]:
    # utils.py:101-107 (pure-python linspace excluding the end point)
    return [start + ((stop - start) / steps) * j for j in range(steps)]


def params_2x(cutoff_fn: str = "cosine") -> Params:
    # aev/_computer.py:550-600, aev/_terms.py:188-207,345-366
    import math

    return make_params(
        7, 5.1, 3.5, 19.7, 12.5, 14.1, linspace(0.8, 5.1, 16), linspace(0.8, 3.5, 8),
        linspace(math.pi / 8, math.pi + math.pi / 8, 4), cutoff_fn,
    )


def params_1x(cutoff_fn: str = "cosine") -> Params:
    # aev/_computer.py:498-548
    import math

    return make_params(
        4, 5.2, 3.5, 16.0, 8.0, 32.0, linspace(0.9, 5.2, 16), linspace(0.9, 3.5, 4),
        linspace(math.pi / 16, math.pi + math.pi / 16, 8), cutoff_fn,
    )


def make_params(S, Rcr, Rca, EtaR, EtaA, Zeta, ShfR, ShfA, ShfZ, cutoff_fn: str = "cosine") -> Params:
    p = Params()
    p.cutoff_kind = {"cosine": 0, "smooth": 1}[cutoff_fn]
    p.S, p.nR, p.nA, p.nZ = S, len(ShfR), len(ShfA), len(ShfZ)
    p.Rcr, p.Rca = float(Rcr), float(Rca)
    p.EtaR, p.EtaA, p.Zeta = f32_consts([EtaR, EtaA, Zeta])
    for dst, src in ((p.ShfR, ShfR), (p.ShfA, ShfA), (p.ShfZ, ShfZ)):
        for k, v in enumerate(f32_consts(src)):
            dst[k] = v
    return p


class Oracle:
    """One precision variant of the oracle library ("f64" checker or "f32" timing port)."""

    def __init__(self, kind: str = "f64") -> None:
        build()
        self.kind = kind
        self.dtype = np.float64 if kind == "f64" else np.float32
        self.lib = C.CDLL(os.path.join(_HERE, f"libani_oracle_{kind}.so"))
        L = self.lib
        vp = C.c_void_p
        L.ani_oracle_real_bytes.restype = C.c_int
        assert L.ani_oracle_real_bytes() == np.dtype(self.dtype).itemsize
        L.ani_oracle_num_threads.restype = C.c_int
        L.ani_oracle_set_threads.argtypes = [C.c_int]
        L.ani_oracle_aev_dim.argtypes = [C.POINTER(Params)]
        L.ani_oracle_aev_dim.restype = C.c_int
        L.ani_oracle_nbrs_brute.restype = vp
        L.ani_oracle_nbrs_brute.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_double]
        L.ani_oracle_nbrs_cell.restype = vp
        L.ani_oracle_nbrs_cell.argtypes = [C.c_int64, vp, vp, vp, vp, C.c_double]
        L.ani_oracle_nbrs_size.restype = C.c_int64
        L.ani_oracle_nbrs_size.argtypes = [vp]
        L.ani_oracle_nbrs_export.argtypes = [vp, vp, vp, vp, vp]
        L.ani_oracle_free_nbrs.argtypes = [vp]
        L.ani_oracle_map_to_central.argtypes = [C.c_int64, vp, vp, vp, vp]
        L.ani_oracle_aev_forward.argtypes = [C.POINTER(Params), vp, vp, vp]
        L.ani_oracle_aev_backward.argtypes = [C.POINTER(Params), vp, vp, vp, vp]
        L.ani_oracle_aev_jvp.argtypes = [C.POINTER(Params), vp, vp, vp, vp]
        L.ani_oracle_mlp_tangent_weight_grads.restype = C.c_double
        L.ani_oracle_mlp_tangent_weight_grads.argtypes = [
            C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp,
            C.c_double if kind == "f64" else C.c_float, vp, vp, vp, vp,
        ]
        L.ani_oracle_aev_backward_virial.argtypes = [C.POINTER(Params), vp, vp, vp, vp, vp]
        L.ani_oracle_mlp.argtypes = [
            C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp,
            C.c_double if kind == "f64" else C.c_float, vp, vp, vp, vp, vp,
        ]
        L.ani_oracle_mlp_weight_grads.argtypes = [
            C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp,
            C.c_double if kind == "f64" else C.c_float, vp, vp, vp, vp,
        ]
        L.ani_oracle_energy_forces.restype = C.c_int
        L.ani_oracle_energy_forces.argtypes = [
            C.POINTER(Params), C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp,
            C.c_double if kind == "f64" else C.c_float, vp, vp, vp, vp, vp,
        ]

    # -- helpers ---------------------------------------------------------------------------
    def num_threads(self) -> int:
        return int(self.lib.ani_oracle_num_threads())

    def set_threads(self, n: int) -> None:
        self.lib.ani_oracle_set_threads(int(n))

    def _r(self, x) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(x), dtype=self.dtype)

    @staticmethod
    def _i32(x) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(x), dtype=np.int32)

    @staticmethod
    def _ptr(a: tp.Optional[np.ndarray]):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def _cell(self, cell, pbc):
        if cell is None or pbc is None or not np.any(np.asarray(pbc)):
            return None, None
        return (
            np.ascontiguousarray(np.asarray(cell), dtype=np.float64),
            np.ascontiguousarray(np.asarray(pbc).astype(np.int32)),
        )

    # -- neighbor lists --------------------------------------------------------------------
    def neighbors(self, species, coords, cutoff, cell=None, pbc=None, cell_list=False):
        """Full neighbor list as (start[n+1], j, d[:,3], r); coords are wrapped first under PBC."""
        species = self._i32(species)
        Cn, A = species.shape
        coords = self._r(coords).reshape(Cn * A, 3)
        cell_, pbc_ = self._cell(cell, pbc)
        if cell_ is not None:
            wrapped = np.empty_like(coords)
            self.lib.ani_oracle_map_to_central(Cn * A, self._ptr(coords), self._ptr(cell_),
                                               self._ptr(pbc_), self._ptr(wrapped))
            coords = wrapped
        if cell_list:
            assert Cn == 1
            h = self.lib.ani_oracle_nbrs_cell(A, self._ptr(species), self._ptr(coords),
                                               self._ptr(cell_), self._ptr(pbc_), float(cutoff))
        else:
            h = self.lib.ani_oracle_nbrs_brute(Cn, A, self._ptr(species), self._ptr(coords),
                                                self._ptr(cell_), self._ptr(pbc_), float(cutoff))
        n = self.lib.ani_oracle_nbrs_size(h)
        start = np.empty(Cn * A + 1, dtype=np.int64)
        j = np.empty(n, dtype=np.int32)
        d = np.empty((n, 3), dtype=self.dtype)
        r = np.empty(n, dtype=self.dtype)
        self.lib.ani_oracle_nbrs_export(h, self._ptr(start), self._ptr(j), self._ptr(d), self._ptr(r))
        self.lib.ani_oracle_free_nbrs(h)
        return start, j, d, r

    # -- networks --------------------------------------------------------------------------
    def mlp(self, species, aev, dims, params, celu_alpha=0.1, n_members=None, want_grad=True,
            want_members=False):
        """dims: [S][nl+1] int; params: flat packed array (see pack_networks)."""
        species = self._i32(species).reshape(-1)
        n = species.shape[0]
        dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int32))
        S, nlp1 = dims.shape
        aev = self._r(aev).reshape(n, dims[0, 0])
        params = self._r(params)
        M = int(n_members)
        ae = np.empty(n, dtype=self.dtype)
        g = np.empty_like(aev) if want_grad else None
        me = np.empty((M, n), dtype=self.dtype) if want_members else None
        self.lib.ani_oracle_mlp(n, S, M, nlp1 - 1, self._ptr(dims), self._ptr(params), celu_alpha,
                                self._ptr(species), self._ptr(aev), self._ptr(ae), self._ptr(g),
                                self._ptr(me))
        return ae, g, me

    def mlp_weight_grads(self, species, aev, g_atom, dims, params, celu_alpha=0.1, n_members=None):
        """d (sum_i g_atom[i] * atomic_e[i]) / d params, in the layout of ``params`` (pack_networks)."""
        species = self._i32(species).reshape(-1)
        n = species.shape[0]
        dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int32))
        S, nlp1 = dims.shape
        aev = self._r(aev).reshape(n, dims[0, 0])
        params = self._r(params)
        g_atom = self._r(g_atom).reshape(n)
        out = np.empty_like(params)
        self.lib.ani_oracle_mlp_weight_grads(n, S, int(n_members), nlp1 - 1, self._ptr(dims), self._ptr(params),
                                             celu_alpha, self._ptr(species), self._ptr(aev), self._ptr(g_atom),
                                             self._ptr(out))
        return out

    def mlp_tangent_weight_grads(self, species, aev, tangent, dims, params, celu_alpha=0.1, n_members=None):
        """(S, dS/d params) for S = sum_i tangent_i . d atomic_e[i]/d aev_i (second-order pass of force training)."""
        species = self._i32(species).reshape(-1)
        n = species.shape[0]
        dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int32))
        S, nlp1 = dims.shape
        aev = self._r(aev).reshape(n, dims[0, 0])
        tangent = self._r(tangent).reshape(n, dims[0, 0])
        params = self._r(params)
        out = np.empty_like(params)
        val = self.lib.ani_oracle_mlp_tangent_weight_grads(
            n, S, int(n_members), nlp1 - 1, self._ptr(dims), self._ptr(params), celu_alpha, self._ptr(species),
            self._ptr(aev), self._ptr(tangent), self._ptr(out))
        return float(val), out

    def aev_jvp(self, p: Params, species, coords, tangent, cell=None, pbc=None, cell_list=False):
        """(aev [C,A,L], J t [C,A,L]) for a coordinate-space direction t [C,A,3] (ani_oracle_aev_jvp)."""
        species = self._i32(species)
        Cn, A = species.shape
        n = Cn * A
        coords = self._r(coords).reshape(n, 3)
        tangent = self._r(tangent).reshape(n, 3)
        cell_, pbc_ = self._cell(cell, pbc)
        if cell_ is not None:
            wrapped = np.empty_like(coords)
            self.lib.ani_oracle_map_to_central(n, self._ptr(coords), self._ptr(cell_), self._ptr(pbc_),
                                               self._ptr(wrapped))
            coords = wrapped
        if cell_list:
            h = self.lib.ani_oracle_nbrs_cell(n, self._ptr(species), self._ptr(coords),
                                               self._ptr(cell_), self._ptr(pbc_), p.Rcr)
        else:
            h = self.lib.ani_oracle_nbrs_brute(Cn, A, self._ptr(species), self._ptr(coords),
                                                self._ptr(cell_), self._ptr(pbc_), p.Rcr)
        L = self.lib.ani_oracle_aev_dim(C.byref(p))
        aev = np.empty((Cn, A, L), dtype=self.dtype)
        out = np.empty((Cn, A, L), dtype=self.dtype)
        self.lib.ani_oracle_aev_forward(C.byref(p), h, self._ptr(species), self._ptr(aev))
        self.lib.ani_oracle_aev_jvp(C.byref(p), h, self._ptr(species), self._ptr(tangent), self._ptr(out))
        self.lib.ani_oracle_free_nbrs(h)
        return aev, out

    # -- AEV only --------------------------------------------------------------------------
    def aev(self, p: Params, species, coords, cell=None, pbc=None, cell_list=False,
            grad_aev=None, want_virial=False):
        """AEVs [C,A,L]; with grad_aev also returns d(sum grad_aev*aev)/d coords [C,A,3] (and, with want_virial,
        the fp64 virial [3,3] of that scalar, see ani_oracle_aev_backward_virial)."""
        species = self._i32(species)
        Cn, A = species.shape
        n = Cn * A
        coords = self._r(coords).reshape(n, 3)
        cell_, pbc_ = self._cell(cell, pbc)
        if cell_ is not None:
            wrapped = np.empty_like(coords)
            self.lib.ani_oracle_map_to_central(n, self._ptr(coords), self._ptr(cell_), self._ptr(pbc_),
                                               self._ptr(wrapped))
            coords = wrapped
        if cell_list:
            h = self.lib.ani_oracle_nbrs_cell(n, self._ptr(species), self._ptr(coords),
                                               self._ptr(cell_), self._ptr(pbc_), p.Rcr)
        else:
            h = self.lib.ani_oracle_nbrs_brute(Cn, A, self._ptr(species), self._ptr(coords),
                                                self._ptr(cell_), self._ptr(pbc_), p.Rcr)
        L = self.lib.ani_oracle_aev_dim(C.byref(p))
        out = np.empty((Cn, A, L), dtype=self.dtype)
        self.lib.ani_oracle_aev_forward(C.byref(p), h, self._ptr(species), self._ptr(out))
        gc = None
        if grad_aev is not None:
            ga = self._r(grad_aev).reshape(n, L)
            gc = np.empty((Cn, A, 3), dtype=self.dtype)
            if want_virial:
                vir = np.zeros((3, 3), dtype=np.float64)
                self.lib.ani_oracle_aev_backward_virial(C.byref(p), h, self._ptr(species), self._ptr(ga),
                                                        self._ptr(gc), self._ptr(vir))
                self.lib.ani_oracle_free_nbrs(h)
                return out, gc, vir
            self.lib.ani_oracle_aev_backward(C.byref(p), h, self._ptr(species), self._ptr(ga),
                                             self._ptr(gc))
        self.lib.ani_oracle_free_nbrs(h)
        return (out, gc) if grad_aev is not None else out

    def virial(self, p: Params, species, coords, dims, params, n_members, cell=None, pbc=None, cell_list=False):
        """Virial [3,3] (Hartree) of the NN energy of all atoms: sum_ij dE/d d_ij (x) d_ij; stress = virial / volume."""
        aev = self.aev(p, species, coords, cell, pbc, cell_list)
        _, g, _ = self.mlp(species, aev, dims, params, n_members=n_members)
        _, _, vir = self.aev(p, species, coords, cell, pbc, cell_list, grad_aev=g, want_virial=True)
        return vir

    # -- whole path ------------------------------------------------------------------------
    def energy_forces(self, p: Params, species, coords, dims, params, n_members, sae=None,
                      cell=None, pbc=None, cell_list=False, celu_alpha=0.1, want_aev=False,
                      want_forces=True):
        species = self._i32(species)
        Cn, A = species.shape
        coords = self._r(coords).reshape(Cn, A, 3)
        cell_, pbc_ = self._cell(cell, pbc)
        dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int32))
        S, nlp1 = dims.shape
        params = self._r(params)
        sae_ = None if sae is None else np.ascontiguousarray(np.asarray(sae, dtype=np.float64))
        L = self.lib.ani_oracle_aev_dim(C.byref(p))
        aev = np.empty((Cn, A, L), dtype=self.dtype) if want_aev else None
        ae = np.empty((Cn, A), dtype=self.dtype)
        em = np.empty(Cn, dtype=np.float64)
        f = np.empty((Cn, A, 3), dtype=self.dtype) if want_forces else None
        rc = self.lib.ani_oracle_energy_forces(
            C.byref(p), Cn, A, self._ptr(species), self._ptr(coords), self._ptr(cell_),
            self._ptr(pbc_), int(cell_list), int(n_members), nlp1 - 1, self._ptr(dims),
            self._ptr(params), celu_alpha, self._ptr(sae_), self._ptr(aev), self._ptr(ae),
            self._ptr(em), self._ptr(f))
        if rc != 0:
            raise RuntimeError(f"ani_oracle_energy_forces failed with {rc}")
        return {"aev": aev, "atomic_energies": ae, "energies": em, "forces": f}


def pack_networks(state: tp.Mapping[str, np.ndarray], symbols: tp.Sequence[str], n_members: int,
                  prefix: str = "potentials.nnp.neural_networks."):
    """Flatten a reference-style state dict (members.{m}.atomics.{Sym}.layers.{l}.*,
    final_layer.*) into the oracle's packed layout.  Returns (dims[S][nl+1], flat float64 params)."""
    dims = []
    for sym in symbols:
        ws = []
        l = 0
        while f"{prefix}members.0.atomics.{sym}.layers.{l}.weight" in state:
            ws.append(np.asarray(state[f"{prefix}members.0.atomics.{sym}.layers.{l}.weight"]).shape)
            l += 1
        fw = np.asarray(state[f"{prefix}members.0.atomics.{sym}.final_layer.weight"]).shape
        dims.append([ws[0][1]] + [w[0] for w in ws] + [fw[0]])
    nl = len(dims[0]) - 1
    flat = []
    for m in range(n_members):
        for sym in symbols:
            for l in range(nl):
                name = f"layers.{l}" if l < nl - 1 else "final_layer"
                base = f"{prefix}members.{m}.atomics.{sym}.{name}."
                flat.append(np.asarray(state[base + "weight"], dtype=np.float64).reshape(-1))
                flat.append(np.asarray(state[base + "bias"], dtype=np.float64).reshape(-1))
    return np.asarray(dims, dtype=np.int32), np.concatenate(flat)
