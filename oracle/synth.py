"""Deterministic synthetic inputs for the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).

All values come from a counter hash, so the CPU oracle, the HIP generator
kernel (nbed_amd/csrc/synth.hip) and the Python host produce the same doubles
bit for bit with no RNG state:

    u(stream, k) = splitmix64(((stream << 48) | k) XOR seed)
    val          = (u >> 11) * 2**-53 * 2 - 1          in [-1, 1)

The reference has no synthetic generator (it always builds real integrals via
PySCF, nbed/driver.py:86-104); these shapes stand in for
``mol.intor('int2e')``, ``get_ovlp()``, ``get_hcore()`` and the DFT embedding
potential (nbed/driver.py:845-852).
"""

from __future__ import annotations

import numpy as np

SEED = 20250829
STREAM_ERI = 0
STREAM_OVLP = 1
STREAM_HCORE = 2
STREAM_VEMB_A = 3
STREAM_VEMB_B = 4
STREAM_MISC = 5

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def val(stream: int, k: np.ndarray, seed: int = SEED) -> np.ndarray:
    """Uniform [-1, 1) double from the counter ``k`` of ``stream``."""
    key = (np.uint64(stream) << np.uint64(48)) | np.asarray(k, dtype=np.uint64)
    u = splitmix64(key ^ np.uint64(seed))
    return (u >> np.uint64(11)).astype(np.float64) * (2.0**-53) * 2.0 - 1.0


def tri(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Index of the unordered pair (a, b) in a packed lower triangle."""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    hi = np.maximum(a, b)
    lo = np.minimum(a, b)
    return hi * (hi + np.uint64(1)) // np.uint64(2) + lo


def eri_canon(p, q, r, s) -> np.ndarray:
    """Index of the 8-fold-unique representative of (pq|rs)."""
    return tri(tri(p, q), tri(r, s))


def eri_scale(nao: int) -> float:
    """Magnitude of the synthetic (pq|rs): uniform(-1,1) * eri_scale(N)."""
    return 1.0 / nao


def eri_block(nao: int, p0: int, p1: int, seed: int = SEED) -> np.ndarray:
    """Dense slab eri[p0:p1, :, :, :] of the 8-fold symmetric synthetic ERI."""
    p = np.arange(p0, p1, dtype=np.uint64)[:, None, None, None]
    q = np.arange(nao, dtype=np.uint64)[None, :, None, None]
    r = np.arange(nao, dtype=np.uint64)[None, None, :, None]
    s = np.arange(nao, dtype=np.uint64)[None, None, None, :]
    return val(STREAM_ERI, eri_canon(p, q, r, s), seed) * eri_scale(nao)


def eri_dense(nao: int, seed: int = SEED) -> np.ndarray:
    """Dense C-order (N,N,N,N) synthetic ERI, chemist notation (pq|rs)."""
    return eri_block(nao, 0, nao, seed)


def sym_matrix(stream: int, n: int, seed: int = SEED) -> np.ndarray:
    """Symmetric n x n matrix with entries val(stream, tri(i, j))."""
    i = np.arange(n, dtype=np.uint64)[:, None]
    j = np.arange(n, dtype=np.uint64)[None, :]
    return val(stream, tri(i, j), seed)


def overlap(nao: int, seed: int = SEED) -> np.ndarray:
    """SPD, well conditioned AO overlap: I + 0.1 sym(val)/sqrt(N), unit diagonal."""
    s = 0.1 * sym_matrix(STREAM_OVLP, nao, seed) / np.sqrt(nao)
    np.fill_diagonal(s, 1.0)
    return s


def hcore(nao: int, seed: int = SEED) -> np.ndarray:
    """Core Hamiltonian with a non-degenerate spectrum: 0.2 sym(val) - diag(0.5 (N-p))."""
    h = 0.2 * sym_matrix(STREAM_HCORE, nao, seed)
    h[np.diag_indices(nao)] -= 0.5 * (nao - np.arange(nao))
    return h


def embedding_potential(nao: int, seed: int = SEED) -> np.ndarray:
    """Small symmetric (2,N,N) stand-in for the DFT embedding potential."""
    va = 0.05 * sym_matrix(STREAM_VEMB_A, nao, seed)
    vb = va + 1e-3 * sym_matrix(STREAM_VEMB_B, nao, seed)
    return np.stack([va, vb])


def general_matrix(stream: int, m: int, n: int, seed: int = SEED) -> np.ndarray:
    """Non-symmetric m x n matrix val(stream, i*n + j) (test operand)."""
    k = np.arange(m * n, dtype=np.uint64).reshape(m, n)
    return val(stream, k, seed)


def lowdin_orthonormal(s: np.ndarray, h: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Eigenpairs of (h, s): returns (eps, C) with C^T S C = I (Loewdin route)."""
    w, u = np.linalg.eigh(s)
    x = (u / np.sqrt(w)) @ u.T
    eps, c = np.linalg.eigh(x @ h @ x)
    return eps, x @ c


def problem(nao: int, nocc: tuple[int, int], n_env: int, seed: int = SEED) -> dict:
    """A full synthetic embedded-SCF problem (without the ERI, which is big).

    Returns S, hcore, V_emb, D_env (2,N,N) built from the ``n_env`` lowest
    S-orthonormal eigenvectors of (hcore, S), and ``nelec`` for the embedded
    (active) system, mirroring what the driver hands to ``huzinaga_scf``
    (nbed/driver.py:582-589): the active electron count is the number of
    active occupied orbitals per spin (nbed/driver.py:262-287).
    """
    s = overlap(nao, seed)
    h = hcore(nao, seed)
    v = embedding_potential(nao, seed)
    _, c = lowdin_orthonormal(s, h)
    c_env = c[:, :n_env]
    d_env = np.stack([c_env @ c_env.T, c_env @ c_env.T])
    nelec = (nocc[0] - n_env, nocc[1] - n_env)
    return {
        "nao": nao,
        "S": s,
        "hcore": h,
        "V_emb": v,
        "D_env": d_env,
        "C_env": c_env,
        "nelec": nelec,
        "nocc_total": tuple(nocc),
        "n_env": n_env,
    }


STREAM_DF = 9


def df_factor(nao: int, l0: int, l1: int, scale: float | None = None, seed: int = SEED) -> np.ndarray:
    """Synthetic three-index factor of the integrals, (pq|rs) ~ sum_L B_L[p][q] B_L[r][s] (the density-fitted form of
    get_veff): B[l - l0][p][q] = scale * val(stream 9, l N(N+1)/2 + tri(p, q)), symmetric in
    (p, q) by construction; scale defaults to 1 / N.  One auxiliary function at a time, element by element."""
    npair = nao * (nao + 1) // 2
    out = np.empty((l1 - l0, nao, nao))
    rows, cols = np.tril_indices(nao)
    flat = rows * (rows + 1) // 2 + cols
    for l in range(l0, l1):
        lower = val(STREAM_DF, np.uint64(l) * np.uint64(npair) + flat.astype(np.uint64), seed)
        m = np.zeros((nao, nao))
        m[rows, cols] = lower
        out[l - l0] = m + np.tril(m, -1).T
    return out * (1.0 / nao if scale is None else scale)
