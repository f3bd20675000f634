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
    if not Path(p).exists():
        build()
    lib = ctypes.CDLL(str(p))
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    lib.jk_ref.argtypes = [dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
    lib.jk_ref.restype = None
    lib.ao2mo_ref.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp,
                              ctypes.c_int, dp, dp]
    lib.ao2mo_ref.restype = None
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
