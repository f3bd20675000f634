"""ctypes loader for the C oracle (oracle/c/jk_ref.c).  TEST INFRASTRUCTURE."""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

CDIR = Path(__file__).resolve().parent / "c"
_lib = None


def build(native_dir: str | None = None) -> Path:
    """Compile the C oracle: portable build in-tree, or -march=native into ``native_dir``."""
    if native_dir is None:
        subprocess.run(["make", "-C", str(CDIR)], check=True, capture_output=True)
        return CDIR / "libjkref.so"
    subprocess.run(["make", "-C", str(CDIR), "native", f"OUT={native_dir}"], check=True, capture_output=True)
    return Path(native_dir) / "libjkref.so"


def load(path: Path | None = None):
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path if path is not None else CDIR / "libjkref.so"
    if not Path(p).exists() or (path is None and any(
            src.stat().st_mtime > Path(p).stat().st_mtime for src in CDIR.glob("*.c"))):
        build()
    lib = ctypes.CDLL(str(p))
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    lib.jk_ref.argtypes = [dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
    lib.jk_ref.restype = None
    lib.ao2mo_ref.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp,
                              ctypes.c_int, dp, dp]
    lib.ao2mo_ref.restype = None
    lib.synth_eri_ref.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, dp]
    lib.synth_eri_ref.restype = None
    lib.jk_synth_sym_ref.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, dp, ctypes.c_int, dp]
    lib.jk_synth_sym_ref.restype = None
    lib.half_transform_rs_ref.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, dp, dp, ctypes.c_int, dp]
    lib.half_transform_rs_ref.restype = None
    if path is None:
        _lib = lib
    return lib


def jk(eri: np.ndarray, dm: np.ndarray, p0: int = 0, p1: int | None = None, lib=None) -> np.ndarray:
    """(1+ndm, p1-p0, N): J of the summed density, K per density (layout of nbx_jk_dense)."""
    lib = lib or load()
    n = dm.shape[-1]
    p1 = n if p1 is None else p1
    dm3 = np.ascontiguousarray(dm.reshape(-1, n, n))
    out = np.empty((1 + dm3.shape[0], p1 - p0, n))
    lib.jk_ref(np.ascontiguousarray(eri), dm3, dm3.shape[0], n, p0, p1, out)
    return out


def ao2mo(eri: np.ndarray, c1, c2, c3, c4, lib=None) -> np.ndarray:
    lib = lib or load()
    n = eri.shape[-1]
    cs = [np.ascontiguousarray(c) for c in (c1, c2, c3, c4)]
    n1, n2, n3, n4 = (c.shape[1] for c in cs)
    out = np.empty((n1, n2, n3, n4))
    work = np.empty(n1 * n**3 + n1 * n2 * n**2 + n1 * n2 * n3 * n)
    lib.ao2mo_ref(np.ascontiguousarray(eri), n, cs[0], n1, cs[1], n2, cs[2], n3, cs[3], n4, out, work)
    return out


SEED = 20250829


def synth_eri(n: int, p0: int = 0, p1: int | None = None, seed: int = SEED, lib=None) -> np.ndarray:
    """Rows [p0,p1) of the dense synthetic (pq|rs): oracle.synth.eri_block in C (seconds at N = 148)."""
    lib = lib or load()
    p1 = n if p1 is None else p1
    out = np.empty((p1 - p0, n, n, n))
    lib.synth_eri_ref(n, p0, p1, seed, out)
    return out


def jk_synth_sym(n: int, dm: np.ndarray, p0: int = 0, p1: int | None = None, seed: int = SEED, lib=None) -> np.ndarray:
    """(1+ndm, N, N) additive symmetric J/K contributions of slab rows [p0,p1) (pairs q <= p and
    their mirror images), integrals generated on the fly: the layout of nbx_jk_synth_sym."""
    lib = lib or load()
    p1 = n if p1 is None else p1
    dm3 = np.ascontiguousarray(dm.reshape(-1, n, n))
    out = np.empty((1 + dm3.shape[0], n, n))
    lib.jk_synth_sym_ref(n, p0, p1, seed, dm3, dm3.shape[0], out)
    return out


def half_transform_rs(n: int, r: int, a: np.ndarray, b: np.ndarray, seed: int = SEED, lib=None) -> np.ndarray:
    """y[k, s] = sum_pq a[k,p] b[k,q] (pq|rs) for s in [0, r]: quarters 1-2 of sampled (i, j) pairs."""
    lib = lib or load()
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    m = a.shape[0]
    assert a.shape == b.shape == (m, n) and m <= 64
    y = np.empty((m, r + 1))
    lib.half_transform_rs_ref(n, seed, r, a, b, m, y)
    return y
