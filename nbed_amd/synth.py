"""Deterministic synthetic inputs for benchmarks and smoke runs.

Counter hash identical to the device generator (csrc/synth_device.h):
    u(stream, k) = splitmix64(((stream << 48) | k) XOR seed);  val = (u >> 11) * 2^-53 * 2 - 1
The big tensor, (pq|rs), is generated on the GPU (``HipBackend.synth_eri``); this module only
builds the O(N^2) operands on the host (hash -> array, no linear algebra) and derives the
environment density with the GPU eigensolver.  The reference has no synthetic generator: these
stand in for ``get_ovlp()``, ``get_hcore()`` and the DFT embedding potential
(nbed/driver.py:845-852).
"""

from __future__ import annotations

import numpy as np

SEED = 20250829
STREAM_OVLP, STREAM_HCORE, STREAM_VEMB_A, STREAM_VEMB_B = 1, 2, 3, 4


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def val(stream: int, k, seed: int = SEED):
    key = (np.uint64(stream) << np.uint64(48)) | np.asarray(k, dtype=np.uint64)
    u = _splitmix64(key ^ np.uint64(seed))
    return (u >> np.uint64(11)).astype(np.float64) * (2.0**-53) * 2.0 - 1.0


def _tri(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    hi, lo = np.maximum(a, b), np.minimum(a, b)
    return hi * (hi + np.uint64(1)) // np.uint64(2) + lo


def sym_matrix(stream: int, n: int, seed: int = SEED):
    i = np.arange(n, dtype=np.uint64)[:, None]
    j = np.arange(n, dtype=np.uint64)[None, :]
    return val(stream, _tri(i, j), seed)


STREAM_DF = 9


def df_scale(nao: int) -> float:
    """Magnitude of the synthetic three-index factor: (pq|pq) ~ naux / (3 N^2) stays of order 1 / N for naux ~ 3 N."""
    return 1.0 / nao


def df_factor(nao: int, l0: int, l1: int, scale: float | None = None, seed: int = SEED):
    """B[l - l0][p][q] = scale * val(stream 9, l N(N+1)/2 + tri(p, q)): what nbx_df_synth generates on the device."""
    npair = np.uint64(nao * (nao + 1) // 2)
    i = np.arange(nao, dtype=np.uint64)[:, None]
    j = np.arange(nao, dtype=np.uint64)[None, :]
    tri = _tri(i, j)
    ls = np.arange(l0, l1, dtype=np.uint64)[:, None, None]
    return (df_scale(nao) if scale is None else scale) * val(STREAM_DF, ls * npair + tri[None], seed)


def overlap(nao: int, seed: int = SEED):
    s = 0.1 * sym_matrix(STREAM_OVLP, nao, seed) / np.sqrt(nao)
    np.fill_diagonal(s, 1.0)
    return s


def hcore(nao: int, seed: int = SEED):
    h = 0.2 * sym_matrix(STREAM_HCORE, nao, seed)
    h[np.diag_indices(nao)] -= 0.5 * (nao - np.arange(nao))
    return h


def embedding_potential(nao: int, seed: int = SEED):
    va = 0.05 * sym_matrix(STREAM_VEMB_A, nao, seed)
    vb = va + 1e-3 * sym_matrix(STREAM_VEMB_B, nao, seed)
    return np.stack([va, vb])


def orthonormal_orbitals(be, s, h):
    """S-orthonormal eigenvectors of (h, S) by Loewdin orthogonalisation on the GPU."""
    s_d, h_d = be.asarray(s), be.asarray(h)
    x = be.sym_pow(s_d, -0.5)
    e, c = be.eigh(be.gemm(be.gemm(x, h_d), x))
    return be.to_host(e), be.to_host(be.gemm(x, c))


def problem(be, nao: int, nocc: tuple[int, int], n_env: int, seed: int = SEED) -> dict:
    """S, hcore, V_emb and the environment density of the n_env lowest orbitals of (hcore, S);
    ``nelec`` is the active electron count handed to the embedded SCF (driver.py:262-287)."""
    s, h, v = overlap(nao, seed), hcore(nao, seed), embedding_potential(nao, seed)
    _, c = orthonormal_orbitals(be, s, h)
    c_env = np.ascontiguousarray(c[:, :n_env])
    d_env_1 = be.to_host(be.gemm(be.asarray(c_env), be.asarray(c_env), "N", "T"))
    return {
        "nao": nao, "S": s, "hcore": h, "V_emb": v, "C": c, "C_env": c_env,
        "D_env": np.stack([d_env_1, d_env_1]),
        "nelec": (nocc[0] - n_env, nocc[1] - n_env), "n_env": n_env,
    }
