"""Loader and ctypes prototypes of libanihip.so (the C ABI declared in include/anihip.h).

The product path has NO CPU fallback: if the library is missing or there is no GPU the calls raise.
``build()`` compiles the HIP sources in-tree with plain ``hipcc --offload-arch=gfx950`` (no hipify, no
torch cpp_extension), so the resulting .so travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import typing as tp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TORCHANI_AMD_LIB") or os.path.join(_HERE, "libanihip.so")
SOURCES = ["api.hip", "nbr.hip", "aev.hip", "aev_generic.hip", "mlp.hip", "mlp_fused.hip", "mlp_prep.hip", "pair.hip", "pack.hip", "train.hip"]
HEADERS = ["anihip_common.h", "train.h", "mlp_fused.h", "mlp_prep.h", os.path.join("..", "..", "include", "anihip.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared"]

MAX_SPECIES = 8
MAX_LAYERS = 4
META_WORDS = 6
STATUS_WORDS = 8
MAX_ANG = 128
MAX_RAD = 256
TABLE_FLOATS = 144
ST_ENTRY_OVERFLOW, ST_ROW_OVERFLOW, ST_GRID_OVERFLOW = 1, 2, 4
MLP_FP32, MLP_F16X3 = 0, 1
BWD_SYMMETRIC, BWD_FIXED_POINT = 1, 2   # flags of anihip_aev_backward
PAIR_PUSH = 1
PAIR_NO_CLAMP = 2
PAIR_XTB, PAIR_ZBL, PAIR_LJ, PAIR_COULOMB = 0, 1, 2, 3
ACT_CELU, ACT_GELU = 0, 1
# anihip_mlp_desc.flags (ANIHIP_MLP_FLAG_*)
MLP_FLAG_NO_FUSED, MLP_FLAG_BIG_TILES, MLP_FLAG_SMALL_TILES, MLP_FLAG_NO_SLAB_MASK, MLP_FLAG_D0_ROWS = 1, 2, 4, 8, 32
MLP_FLAG_FUSED_L0B, MLP_FLAG_NO_FUSED_L0B = 512, 1024
MLP_FLAG_SHAPED = 4096   # one fused launch per species with compile-time network widths (include/anihip.h)
MLP_FLAG_BWD_TWO_PRODUCTS = 2048   # off by default: two-product backward GEMMs of the large-system path (include/anihip.h)
ABI_VERSION = 12
REPACK_FUSED_ONLY = 1


class AevParams(C.Structure):
    _fields_ = [
        ("num_species", C.c_int32),
        ("n_shf_r", C.c_int32),
        ("n_shf_a", C.c_int32),
        ("n_shf_z", C.c_int32),
        ("Rcr", C.c_float),
        ("Rca", C.c_float),
        ("EtaR", C.c_float),
        ("EtaA", C.c_float),
        ("Zeta", C.c_float),
        ("cutoff_kind", C.c_int32),
        ("flags", C.c_int32),   # ANIHIP_AEV_*: set by anihip_aev_table_pack
    ]


CUTOFF_KINDS = {"cosine": 0, "smooth": 1}  # ANIHIP_CUTOFF_*


class SpeciesNet(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32),
        ("dims", C.c_int32 * (MAX_LAYERS + 1)),
        ("w", C.c_void_p * MAX_LAYERS),
        ("wt", C.c_void_p * MAX_LAYERS),
        ("bias", C.c_void_p * MAX_LAYERS),
        ("wh", C.c_void_p * MAX_LAYERS),
        ("wth", C.c_void_p * MAX_LAYERS),
        ("wh_scale", C.c_float * MAX_LAYERS),
        ("whf", C.c_void_p * MAX_LAYERS),
        ("wthf", C.c_void_p * MAX_LAYERS),
        ("fused_bounds", C.c_void_p),
    ]


class SpeciesGrads(C.Structure):
    _fields_ = [
        ("gw", C.c_void_p * MAX_LAYERS),
        ("gbias", C.c_void_p * MAX_LAYERS),
        ("member_stride", C.c_int64),
        ("accumulate", C.c_int32),
    ]


class D3Params(C.Structure):
    """anihip_d3_params (include/anihip.h)."""
    _fields_ = [("s6", C.c_float), ("s8", C.c_float), ("a1", C.c_float), ("a2", C.c_float),
                ("cov_radius_bohr", C.c_float * 8), ("sqrt_q", C.c_float * 8)]


class MlpDesc(C.Structure):
    _fields_ = [
        ("num_species", C.c_int32),
        ("n_members", C.c_int32),
        ("aev_len", C.c_int32),
        ("celu_alpha", C.c_float),
        ("precision", C.c_int32),
        ("aev_radial_len", C.c_int32),
        ("flags", C.c_int32),
        ("activation", C.c_int32),
        ("net", SpeciesNet * MAX_SPECIES),
    ]


class MlpShape(C.Structure):
    """anihip_mlp_shape (include/anihip.h): what anihip_mlp_pack needs to know about the networks."""
    _fields_ = [("n_members", C.c_int32), ("num_species", C.c_int32), ("n_layers", C.c_int32), ("aev_len", C.c_int32),
                ("aev_radial_len", C.c_int32), ("precision", C.c_int32), ("activation", C.c_int32),
                ("celu_alpha", C.c_float), ("out_dims", (C.c_int32 * MAX_LAYERS) * MAX_SPECIES)]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_HERE, "csrc", s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libanihip.so for gfx950 if it is missing or older than its sources."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libanihip.so")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(_HERE, "csrc", s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib: tp.Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """The loaded library.  torch must be imported first so that both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first; ours resolves to the same soname)

    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine is not built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (there is no CPU fallback)."
        )
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.anihip_last_error.restype = C.c_char_p
    L.anihip_abi_version.restype = C.c_int
    L.anihip_aev_table_pack.argtypes = [C.POINTER(AevParams), vp, vp, vp, vp]
    L.anihip_nbr_workspace_bytes.restype = sz
    L.anihip_nbr_workspace_bytes.argtypes = [i64, i64]
    L.anihip_nbr_build_batch.argtypes = [vp, C.POINTER(AevParams), i32, i32, vp, vp, vp, i32, i64, i64, vp,
                                         sz, vp, vp, i64, vp]
    L.anihip_nbr_build_cell.argtypes = [vp, C.POINTER(AevParams), i64, vp, vp, vp, i32, i64, i64, i64, vp,
                                        sz, vp, vp, i64, vp]
    L.anihip_nbr_half_workspace_bytes.restype = sz
    L.anihip_nbr_half_workspace_bytes.argtypes = [i64]
    L.anihip_nbr_from_half.argtypes = [vp, C.POINTER(AevParams), i64, vp, i64, vp, vp, i64, i64, vp, sz, vp, vp,
                                       i64, vp]
    L.anihip_nbr_rows_to_half_workspace_bytes.restype = sz
    L.anihip_nbr_rows_to_half_workspace_bytes.argtypes = [i64]
    L.anihip_nbr_rows_to_half.restype = C.c_int
    L.anihip_nbr_rows_to_half.argtypes = [vp, i64, i64, i64, vp, vp, vp, sz, i64, vp, vp, vp, vp]
    L.anihip_nbr_from_full.argtypes = [vp, C.POINTER(AevParams), i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp]
    L.anihip_nbr_refresh.argtypes = [vp, C.POINTER(AevParams), i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.anihip_aev_forward.argtypes = [vp, C.POINTER(AevParams), vp, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    L.anihip_aev_forward_update.argtypes = [vp, C.POINTER(AevParams), vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.anihip_aev_backward.argtypes = [vp, C.POINTER(AevParams), vp, i64, i64, i64, vp, vp, vp, vp, vp, i32, vp, vp]
    L.anihip_aev_jvp.argtypes = [vp, C.POINTER(AevParams), vp, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    L.anihip_aev_backward_virial.argtypes = [vp, C.POINTER(AevParams), vp, i64, i64, i64, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.anihip_mlp_workspace_bytes.restype = sz
    L.anihip_mlp_workspace_bytes.argtypes = [C.POINTER(MlpDesc), i64]
    L.anihip_mlp_forward_backward_workspace_bytes.restype = sz
    L.anihip_mlp_forward_backward_workspace_bytes.argtypes = [C.POINTER(MlpDesc), i64, C.c_int32]
    L.anihip_mlp_pack_bytes.restype = sz
    L.anihip_mlp_pack_bytes.argtypes = [C.POINTER(MlpShape)]
    L.anihip_mlp_pack.restype = C.c_int
    L.anihip_mlp_pack.argtypes = [vp, C.POINTER(MlpShape), vp, vp, i32, vp, sz, i32, C.POINTER(MlpDesc)]
    L.anihip_mlp_forward_backward.argtypes = [vp, C.POINTER(MlpDesc), i64, i64, i64, vp, vp, vp, vp, sz, vp,
                                              vp, vp]
    L.anihip_mlp_train_workspace_bytes.restype = sz
    L.anihip_mlp_train_workspace_bytes.argtypes = [C.POINTER(MlpDesc), i64]
    L.anihip_mlp_weight_grads.argtypes = [vp, C.POINTER(MlpDesc), i64, i64, i64, vp, vp, vp, vp, sz,
                                          C.POINTER(SpeciesGrads), vp, vp, i32]
    L.anihip_mlp_tangent_workspace_bytes.restype = sz
    L.anihip_mlp_tangent_workspace_bytes.argtypes = [C.POINTER(MlpDesc), i64]
    L.anihip_mlp_tangent_weight_grads.argtypes = [vp, C.POINTER(MlpDesc), i64, i64, i64, vp, vp, vp, vp, sz,
                                                  C.POINTER(SpeciesGrads), vp]
    L.anihip_mlp_train_forward.argtypes = [vp, C.POINTER(MlpDesc), i64, i64, i64, vp, vp, vp, sz, vp]
    L.anihip_mlp_repack.argtypes = [vp, C.POINTER(MlpDesc), vp, vp, vp, i32]
    L.anihip_adam_step.argtypes = [vp, vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, vp, i32]
    L.anihip_adam_step.restype = C.c_int
    L.anihip_energy_reduce.argtypes = [vp, i32, i32, i64, i64, vp, vp, vp, vp]
    L.anihip_energy_forces_finish.argtypes = [vp, i32, i32, i64, i64, vp, vp, vp, vp, vp, i64]
    L.anihip_energy_forces_finish.restype = C.c_int
    L.anihip_pair_xtb_repulsion.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, C.c_float, i32, i32, vp, vp, vp]
    L.anihip_pair_xtb_repulsion.restype = C.c_int
    L.anihip_pair_analytic.argtypes = [vp, i32, i64, i64, i64, vp, vp, vp, vp, vp, C.c_float, i32, i32, vp, vp, vp]
    L.anihip_pair_analytic.restype = C.c_int
    L.anihip_pair_d3.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, C.POINTER(D3Params), C.c_float, i32, vp, vp, vp, vp, vp]
    L.anihip_pair_d3.restype = C.c_int
    for name in ("anihip_aev_table_pack", "anihip_nbr_build_batch", "anihip_nbr_build_cell", "anihip_nbr_from_half",
                 "anihip_nbr_from_full", "anihip_nbr_refresh",
                 "anihip_aev_forward", "anihip_aev_forward_update", "anihip_aev_backward", "anihip_aev_backward_virial", "anihip_aev_jvp", "anihip_mlp_forward_backward",
                 "anihip_mlp_weight_grads", "anihip_mlp_tangent_weight_grads", "anihip_mlp_train_forward", "anihip_mlp_repack", "anihip_energy_reduce"):
        getattr(L, name).restype = C.c_int
    if L.anihip_abi_version() != ABI_VERSION:
        raise RuntimeError("libanihip.so ABI version mismatch: rebuild it")
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "anihip_last_error", "anihip_abi_version", "anihip_aev_table_pack", "anihip_nbr_workspace_bytes",
    "anihip_nbr_build_batch", "anihip_nbr_build_cell", "anihip_nbr_half_workspace_bytes", "anihip_nbr_from_half", "anihip_nbr_from_full",
    "anihip_nbr_refresh", "anihip_aev_forward", "anihip_aev_forward_update", "anihip_aev_backward", "anihip_aev_backward_virial", "anihip_aev_jvp",
    "anihip_mlp_workspace_bytes", "anihip_mlp_forward_backward_workspace_bytes", "anihip_mlp_forward_backward", "anihip_mlp_train_workspace_bytes",
    "anihip_mlp_weight_grads", "anihip_mlp_train_forward", "anihip_mlp_repack", "anihip_adam_step", "anihip_energy_reduce",
    "anihip_mlp_tangent_workspace_bytes", "anihip_mlp_tangent_weight_grads", "anihip_pair_xtb_repulsion",
    "anihip_pair_d3", "anihip_pair_analytic", "anihip_energy_forces_finish", "anihip_mlp_pack_bytes", "anihip_mlp_pack",
    "anihip_nbr_rows_to_half_workspace_bytes", "anihip_nbr_rows_to_half",
]


def check(rc: int) -> None:
    """Map a non-zero status of the C ABI to RuntimeError (the reference raises c10::Error ->
    RuntimeError from TORCH_CHECK, csrc/aev.cu:1693-1710)."""
    if rc != 0:
        raise RuntimeError("libanihip: " + lib().anihip_last_error().decode())
