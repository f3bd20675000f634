"""ctypes binding of libnbx (include/nbx.h) -- the thin FFI layer.

Only plain pointers and sizes cross this boundary.  The product path fails
loudly (``NbxUnavailableError``) when the shared library is missing; there is no
CPU fallback anywhere in ``nbed_amd``.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char, c_char_p, c_double, c_int, c_int64, c_size_t, c_uint64, c_void_p
from pathlib import Path

LIB_NAME = "libnbx.so"
LIB_PATH = Path(__file__).resolve().parent / LIB_NAME

NBX_VERSION = 3  # the include/nbx.h this table mirrors (buffer sizes behind the entry points: see the header)
NBX_OK = 0
NBX_E_INVALID = -1
NBX_E_HIP = -2
NBX_E_NOMEM = -3
NBX_E_NOCONV = -4
NBX_E_UNSUPPORTED = -5

HUZ_JK_PACKED, HUZ_JK_SYM = 0, 1
XC_CODES = {"slater": 0, "lda": 1, "lda,vwn_rpa": 1, "lda,vwn": 2, "lda,vwn5": 2, "svwn": 2, "b3lyp": 3}

PROF_JK_DENSE, PROF_AO2MO_Q1, PROF_AO2MO, PROF_EIGH, PROF_SVD, PROF_GEMM = range(6)


class NbxError(RuntimeError):
    """A libnbx entry point returned an error code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libnbx error {code}: {message}")
        self.code = code


class NbxUnavailableError(RuntimeError):
    """libnbx.so is not built or no MI355X is visible: the HIP path cannot run."""


class HuzState(ctypes.Structure):
    """``nbx_huz_state`` of include/nbx.h: the device pointers one SCF keeps for ``nbx_huz_cycle``."""

    _fields_ = [
        ("nao", c_int64), ("nocc_a", c_int64), ("nocc_b", c_int64),
        ("d_packed", c_void_p), ("d_hv", c_void_p), ("d_ds", c_void_p), ("d_sb", c_void_p), ("d_x", c_void_p),
        ("d_dts", c_void_p), ("d_jk", c_void_p), ("d_fock", c_void_p), ("d_vhf", c_void_p), ("d_fock2", c_void_p),
        ("d_tmp", c_void_p), ("d_fo", c_void_p),
        ("d_jk_work", c_void_p), ("jk_work_bytes", c_size_t),
        ("d_eig_work", c_void_p), ("eig_work_bytes", c_size_t),
        ("d_geig_work", c_void_p), ("geig_work_bytes", c_size_t),
        ("diis_space", c_int64), ("d_diis_xs", c_void_p), ("d_diis_es", c_void_p), ("d_diis_h", c_void_p),
        ("d_diis_coef", c_void_p), ("d_diis_xprev", c_void_p),
        ("jk_kind", c_int64), ("jk_p0", c_int64), ("jk_p1", c_int64), ("d_eri", c_void_p),
    ]


# name -> (restype, argtypes); mirrors include/nbx.h one to one
_P = c_void_p
SIGNATURES = {
    "nbx_version": (c_int, []),
    "nbx_experimental": (c_int, []),
    "nbx_last_error": (c_char_p, []),
    "nbx_device_count": (c_int, [POINTER(c_int)]),
    "nbx_ctx_create": (c_int, [c_int, _P, c_int, POINTER(_P)]),
    "nbx_ctx_destroy": (c_int, [_P]),
    "nbx_ctx_set_stream": (c_int, [_P, _P]),
    "nbx_sync": (c_int, [_P]),
    "nbx_malloc": (c_int, [_P, c_size_t, POINTER(_P)]),
    "nbx_free": (c_int, [_P, _P]),
    "nbx_memcpy_h2d": (c_int, [_P, _P, _P, c_size_t]),
    "nbx_memcpy_d2h": (c_int, [_P, _P, _P, c_size_t]),
    "nbx_gather_to_host": (c_int, [_P, c_int64, _P, _P, _P, c_int]),
    "nbx_jk_df_worksize": (c_size_t, [c_int64, c_int64, c_int64]),
    "nbx_jk_df": (c_int, [_P, c_int64, c_int64, _P, c_int64, _P, POINTER(c_int64), _P, _P, c_size_t]),
    "nbx_df_synth": (c_int, [_P, c_int64, c_int64, c_int64, c_uint64, c_double, _P]),
    "nbx_memcpy_d2d": (c_int, [_P, _P, _P, c_size_t]),
    "nbx_memset": (c_int, [_P, _P, c_int, c_size_t]),
    "nbx_profile_enable": (c_int, [_P, c_int]),
    "nbx_profile_sample": (c_int, [_P, c_int]),
    "nbx_debug_fill_lds": (c_int, [_P, c_double]),
    "nbx_profile_read": (c_int, [_P, c_int, POINTER(c_double), POINTER(c_int64)]),
    "nbx_profile_reset": (c_int, [_P]),
    "nbx_synth_eri": (c_int, [_P, c_int64, c_int64, c_int64, c_uint64, _P]),
    "nbx_jk_dense_worksize": (c_size_t, [c_int64, c_int64, c_int64]),
    "nbx_jk_dense": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P, c_size_t]),
    "nbx_jk_dense_sym_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "nbx_jk_packed_supported": (c_int, [c_int64]),
    "nbx_jk_packed_fold": (c_int, [c_int64]),
    "nbx_eri_packed_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "nbx_eri_pack": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    "nbx_jk_packed_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "nbx_huzinaga_fused": (c_int, [_P, c_int64, c_int64, _P, _P, c_double, _P, _P]),
    "nbx_jk_packed_fock": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "nbx_jk_dts_bytes": (c_size_t, [c_int64]),
    "nbx_jk_dts_init": (c_int, [_P, c_int64, _P]),
    "nbx_jk_packed": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P, c_size_t]),
    "nbx_jk_dense_sym": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P, c_size_t]),
    "nbx_jk_synth_sym_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "nbx_jk_synth_sym": (c_int, [_P, c_int64, c_int64, c_int64, c_uint64, _P, c_int64, _P, _P, c_size_t]),
    "nbx_jk_synth": (c_int, [_P, c_int64, c_int64, c_int64, c_uint64, _P, c_int64, _P, _P, c_size_t]),
    "nbx_gemm": (c_int, [_P, c_char, c_char, c_int64, c_int64, c_int64, c_double, _P, c_int64, c_int64,
                         _P, c_int64, c_int64, c_double, _P, c_int64, c_int64, c_int64]),
    "nbx_fock_uhf": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P, _P]),
    "nbx_huzinaga_sym": (c_int, [_P, c_int64, c_int64, _P, c_double, _P, _P]),
    "nbx_trace_prod": (c_int, [_P, c_int64, c_int64, _P, _P, POINTER(c_double)]),
    "nbx_huz_cycle_scalars": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P, _P, _P, POINTER(c_double)]),
    "nbx_huz_cycle_scalars_dev": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int64]),
    "nbx_huz_cycle_scalars_dts": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int64, _P]),
    "nbx_diis_coef_doubles": (c_size_t, [c_int64]),
    "nbx_diis_update": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P]),
    "nbx_diis_update_err": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "nbx_vo_sumsq": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P]),
    "nbx_axpby": (c_int, [_P, c_int64, c_double, _P, c_double, _P]),
    "nbx_lincomb": (c_int, [_P, c_int64, c_int64, POINTER(c_double), _P, c_int64, _P]),
    "nbx_dots": (c_int, [_P, c_int64, c_int64, _P, _P, c_int64, POINTER(c_double)]),
    "nbx_transpose": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    "nbx_scale_cols": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    "nbx_eigh_worksize": (c_size_t, [c_int64, c_int64]),
    "nbx_eigh": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, c_size_t]),
    "nbx_eigh_warm": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, _P, c_size_t]),
    "nbx_geig_refine_worksize": (c_size_t, [c_int64, c_int64]),
    "nbx_geig_refine": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_size_t, c_int]),
    "nbx_eigh_warm_ex": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, _P, c_size_t, c_int]),
    "nbx_eigh_status_offset": (c_size_t, [c_int64, c_int64]),
    "nbx_eigh_approx_worksize": (c_size_t, [c_int64, c_int64]),
    "nbx_eigh_approx": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "nbx_eigh_status": (c_int, [_P, c_int64, c_int64, _P, POINTER(c_int)]),
    "nbx_sym_pow_worksize": (c_size_t, [c_int64]),
    "nbx_sym_pow": (c_int, [_P, c_int64, _P, c_double, _P, _P, c_size_t]),
    "nbx_sym_pow_ns_worksize": (c_size_t, [c_int64]),
    "nbx_sym_pow_ns": (c_int, [_P, c_int64, _P, c_double, c_double, _P, _P, c_size_t, c_int, c_int, POINTER(c_int)]),
    "nbx_svd_worksize": (c_size_t, [c_int64, c_int64]),
    "nbx_svd_right": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, c_size_t]),
    "nbx_svd_status": (c_int, [_P, c_int64, c_int64, _P, POINTER(c_int)]),
    "nbx_ao2mo_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "nbx_ao2mo": (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P,
                          c_int64, _P, _P, c_size_t]),
    "nbx_ao2mo_pair_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "nbx_ao2mo_pair": (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64,
                               _P, _P, c_int64, _P, c_int64, _P, _P, c_size_t]),
    "nbx_ao2mo_pair_sym_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "nbx_ao2mo_pair_sym": (c_int, [_P, c_int64, _P, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64,
                                   _P, _P, c_size_t]),
    "nbx_eri_rs_bytes": (c_size_t, [c_int64]),
    "nbx_eri_pack_rs": (c_int, [_P, c_int64, _P, _P]),
    "nbx_ao2mo_pair_sym_rs_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "nbx_ao2mo_pair_sym_rs": (c_int, [_P, c_int64, _P, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P,
                                      c_int64, _P, _P, c_size_t]),
    "nbx_ao2mo_synth_pair_worksize": (c_size_t, [c_int64] * 7),
    "nbx_ao2mo_synth_pair": (c_int, [_P, c_int64, c_uint64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64,
                                     _P, c_int64, _P, _P, c_int64, _P, c_int64, _P, _P, c_size_t]),
    "nbx_ao2mo_synth_worksize": (c_size_t, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "nbx_ao2mo_synth": (c_int, [_P, c_int64, c_uint64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P,
                                c_int64, _P, _P, c_size_t]),
    "nbx_chem_to_phys": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P, _P]),
    "nbx_spinorb_scatter_h1": (c_int, [_P, c_int64, _P, c_double, _P]),
    "nbx_spinorb_scatter_range": (c_int, [_P, c_int64, _P, c_double, c_double, c_int64, c_int64, _P]),
    "nbx_spinorb_scatter": (c_int, [_P, c_int64, _P, _P, c_double, c_double, _P, _P]),
    "nbx_threshold_scale": (c_int, [_P, c_int64, c_double, c_double, _P]),
    "nbx_becke_share": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int64, _P]),
    "nbx_eval_ao": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int64, _P, _P]),
    "nbx_xc_rho": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "nbx_xc_functional_worksize": (c_size_t, [c_int64]),
    "nbx_xc_functional": (c_int, [_P, c_int, c_int64, _P, _P, _P, c_double, _P, _P, _P, _P, c_size_t]),
    "nbx_xc_vmat_worksize": (c_size_t, [c_int64, c_int64]),
    "nbx_xc_vmat": (c_int, [_P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, c_size_t]),
    "nbx_purify_worksize": (c_size_t, [c_int64, c_int64]),
    "nbx_purify": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_size_t, c_int, _P]),
    "nbx_host_1e": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, _P, _P, _P]),
    "nbx_host_eri": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, c_double, c_int, _P]),
    "nbx_huz_cycle": (c_int, [_P, POINTER(HuzState), _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                              c_int, _P, _P]),
    "nbx_huz_cycle_jk": (c_int, [_P, POINTER(HuzState), _P]),
    "nbx_mu_cycle": (c_int, [_P, POINTER(HuzState), _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                             c_int, _P, _P]),
    "nbx_mu_cycle_solve": (c_int, [_P, POINTER(HuzState), _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                   _P]),
    "nbx_mu_cycle_fock": (c_int, [_P, POINTER(HuzState), _P, _P, _P, _P, _P, c_int, _P, _P]),
    "nbx_mu_cycle_fock_post": (c_int, [_P, POINTER(HuzState), _P, _P, _P, _P, _P, c_int, _P, _P]),
    "nbx_huz_cycle_post": (c_int, [_P, POINTER(HuzState), _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                   _P, _P]),
}

_lib = None


def load_library(path: os.PathLike | None = None) -> ctypes.CDLL:
    """dlopen libnbx.so and attach the prototypes of include/nbx.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None and os.environ.get("NBX_LIB"):  # an alternative build of the library (A/B measurements)
        path = os.environ["NBX_LIB"]
    p = Path(path) if path is not None else LIB_PATH
    if not p.exists():
        raise NbxUnavailableError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); nbed_amd has no CPU fallback."
        )
    try:
        # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's): load torch first so
        # that libnbx binds to the one HIP runtime already in the process.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less FFI users get the system runtime
        pass
    lib = ctypes.CDLL(str(p), mode=ctypes.RTLD_GLOBAL)
    lib.nbx_version.restype = c_int
    if lib.nbx_version() != NBX_VERSION:
        raise NbxUnavailableError(
            f"{p} is ABI version {lib.nbx_version()}, this package binds version {NBX_VERSION} (include/nbx.h): "
            "rebuild it (`make -C nbed_amd/csrc`) -- buffer sizes behind several entry points differ between versions"
        )
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib


def check(lib: ctypes.CDLL, rc: int) -> None:
    if rc != NBX_OK:
        msg = lib.nbx_last_error()
        raise NbxError(rc, msg.decode() if msg else "")
