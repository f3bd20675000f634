"""Oracle: a minimal Gaussian-integral engine (s, p and spherical d, f shells) for real-molecule KATs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference obtains S, T, V_nuc and (pq|rs)
from PySCF/libcint (``gto.Mole.intor``; nbed/driver.py:86-104 builds the molecule).  libcint is
not part of /root/reference, so this module restates the textbook McMurchie-Davidson scheme
(Helgaker, Jorgensen, Olsen, ch. 9) for contracted Cartesian s/p Gaussians -- enough for
STO-3G water, the system of the reference's DFT-free known-answer test
(tests/test_driver.py:52-61: global UHF e_tot, e_nuc, energy_elec).

Conventions follow PySCF so that the numbers are comparable: Bohr radius 0.52917721092 A
(pyscf.data.nist.BOHR), AO order per atom = shells in basis order, p functions as (x, y, z),
normalised contracted functions.

``SphericalBasis`` (further down) adds d and f shells the way libcint defines them: real solid harmonics
normalised on the sphere (m = -l .. l; d: xy, yz, z^2, xz, x^2 - y^2) times a radial contraction whose
primitives are normalised in closed form (gamma functions), each AO a short list of weighted
Cartesian monomials fed to the same primitive recursions.  The product (nbed_amd/integrals.py)
normalises its spherical combinations NUMERICALLY from the shell's own overlap and evaluates
shell pairs vectorised -- two routes to the same numbers.  The basis-set tables themselves are data
(typed once); they are pinned by literature Hartree-Fock energies in tests/test_host_integrals.py.
"""

from __future__ import annotations

import math

import numpy as np
from scipy.special import hyp1f1

BOHR = 0.52917721092

# STO-3G (EMSL / PySCF "sto-3g"): (angular momentum, exponents, contraction coefficients)
_STO3G_CORE = (0.15432897, 0.53532814, 0.44463454)
_STO3G_2S = (-0.09996723, 0.39951283, 0.70011547)
_STO3G_2P = (0.15591627, 0.60768372, 0.39195739)
STO3G = {
    "H": [(0, (3.42525091, 0.62391373, 0.16885540), _STO3G_CORE)],
    "O": [
        (0, (130.7093200, 23.8088610, 6.4436083), _STO3G_CORE),
        (0, (5.0331513, 1.1695961, 0.3803890), _STO3G_2S),
        (1, (5.0331513, 1.1695961, 0.3803890), _STO3G_2P),
    ],
}
CHARGE = {"H": 1, "O": 8}


def parse_xyz(xyz: str):
    """Raw xyz text ('<n>\\n<comment>\\n<sym> x y z ...', Angstrom) -> [(symbol, np.array bohr)]."""
    lines = [ln for ln in xyz.strip().splitlines()[2:] if ln.strip()]
    atoms = []
    for ln in lines:
        sym, x, y, z = ln.split()[:4]
        atoms.append((sym.capitalize(), np.array([float(x), float(y), float(z)]) / BOHR))
    return atoms


def _double_factorial(n: int) -> int:
    return 1 if n <= 0 else n * _double_factorial(n - 2)


class Basis:
    """Contracted Cartesian Gaussians of a molecule."""

    def __init__(self, atoms, basis=STO3G):
        self.atoms = atoms
        self.funcs = []  # (center, (l,m,n), exponents, normalised coefficients)
        self.ao_slices = []
        for iat, (sym, pos) in enumerate(atoms):
            start = len(self.funcs)
            for ang, exps, coefs in basis[sym]:
                lmns = [(0, 0, 0)] if ang == 0 else [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
                for lmn in lmns:
                    self.funcs.append((pos, lmn, np.array(exps), self._normalise(lmn, np.array(exps), np.array(coefs))))
            self.ao_slices.append([iat, iat + 1, start, len(self.funcs)])
        self.nao = len(self.funcs)

    @staticmethod
    def _normalise(lmn, exps, coefs):
        l, m, n = lmn
        L = l + m + n
        # primitive norms
        norm = np.sqrt(
            (2 ** (2 * L + 1.5)) * exps ** (L + 1.5)
            / (_double_factorial(2 * l - 1) * _double_factorial(2 * m - 1) * _double_factorial(2 * n - 1) * math.pi**1.5)
        )
        c = coefs * norm
        # contracted norm
        pref = math.pi**1.5 * _double_factorial(2 * l - 1) * _double_factorial(2 * m - 1) * _double_factorial(2 * n - 1) / 2.0**L
        s = 0.0
        for ci, ai in zip(c, exps):
            for cj, aj in zip(c, exps):
                s += ci * cj / (ai + aj) ** (L + 1.5)
        return c / math.sqrt(pref * s)


def _E(i, j, t, qx, a, b):
    """Hermite expansion coefficient E_t^{ij} (1-D), McMurchie-Davidson recursion."""
    p = a + b
    q = a * b / p
    if t < 0 or t > i + j:
        return 0.0
    if i == j == t == 0:
        return math.exp(-q * qx * qx)
    if j == 0:
        return (1 / (2 * p)) * _E(i - 1, j, t - 1, qx, a, b) - (q * qx / a) * _E(i - 1, j, t, qx, a, b) + (t + 1) * _E(
            i - 1, j, t + 1, qx, a, b)
    return (1 / (2 * p)) * _E(i, j - 1, t - 1, qx, a, b) + (q * qx / b) * _E(i, j - 1, t, qx, a, b) + (t + 1) * _E(
        i, j - 1, t + 1, qx, a, b)


def _boys(n, x):
    return hyp1f1(n + 0.5, n + 1.5, -x) / (2.0 * n + 1.0)


def _R(t, u, v, n, p, pcx, pcy, pcz, rpc2):
    """Hermite Coulomb integral R^n_{tuv}."""
    if t == u == v == 0:
        return (-2 * p) ** n * _boys(n, p * rpc2)
    if t < 0 or u < 0 or v < 0:
        return 0.0
    if t > 0:
        val = pcx * _R(t - 1, u, v, n + 1, p, pcx, pcy, pcz, rpc2)
        if t > 1:
            val += (t - 1) * _R(t - 2, u, v, n + 1, p, pcx, pcy, pcz, rpc2)
        return val
    if u > 0:
        val = pcy * _R(t, u - 1, v, n + 1, p, pcx, pcy, pcz, rpc2)
        if u > 1:
            val += (u - 1) * _R(t, u - 2, v, n + 1, p, pcx, pcy, pcz, rpc2)
        return val
    val = pcz * _R(t, u, v - 1, n + 1, p, pcx, pcy, pcz, rpc2)
    if v > 1:
        val += (v - 1) * _R(t, u, v - 2, n + 1, p, pcx, pcy, pcz, rpc2)
    return val


def _overlap_prim(a, lmn1, A, b, lmn2, B):
    s = 1.0
    for d in range(3):
        s *= _E(lmn1[d], lmn2[d], 0, A[d] - B[d], a, b)
    return s * (math.pi / (a + b)) ** 1.5


def _kinetic_prim(a, lmn1, A, b, lmn2, B):
    l2, m2, n2 = lmn2
    term0 = b * (2 * (l2 + m2 + n2) + 3) * _overlap_prim(a, lmn1, A, b, lmn2, B)
    term1 = -2 * b**2 * (
        _overlap_prim(a, lmn1, A, b, (l2 + 2, m2, n2), B)
        + _overlap_prim(a, lmn1, A, b, (l2, m2 + 2, n2), B)
        + _overlap_prim(a, lmn1, A, b, (l2, m2, n2 + 2), B)
    )
    term2 = -0.5 * (
        l2 * (l2 - 1) * _overlap_prim(a, lmn1, A, b, (l2 - 2, m2, n2), B)
        + m2 * (m2 - 1) * _overlap_prim(a, lmn1, A, b, (l2, m2 - 2, n2), B)
        + n2 * (n2 - 1) * _overlap_prim(a, lmn1, A, b, (l2, m2, n2 - 2), B)
    )
    return term0 + term1 + term2


def _nuclear_prim(a, lmn1, A, b, lmn2, B, C):
    p = a + b
    P = (a * A + b * B) / p
    pc = P - C
    rpc2 = float(pc @ pc)
    val = 0.0
    for t in range(lmn1[0] + lmn2[0] + 1):
        ex = _E(lmn1[0], lmn2[0], t, A[0] - B[0], a, b)
        for u in range(lmn1[1] + lmn2[1] + 1):
            ey = _E(lmn1[1], lmn2[1], u, A[1] - B[1], a, b)
            for v in range(lmn1[2] + lmn2[2] + 1):
                ez = _E(lmn1[2], lmn2[2], v, A[2] - B[2], a, b)
                val += ex * ey * ez * _R(t, u, v, 0, p, pc[0], pc[1], pc[2], rpc2)
    return val * 2 * math.pi / p


def _eri_prim(a, lmn1, A, b, lmn2, B, c, lmn3, C, d, lmn4, D):
    p, q = a + b, c + d
    alpha = p * q / (p + q)
    P = (a * A + b * B) / p
    Q = (c * C + d * D) / q
    pq = P - Q
    rpq2 = float(pq @ pq)
    e1 = [[_E(lmn1[k], lmn2[k], t, A[k] - B[k], a, b) for t in range(lmn1[k] + lmn2[k] + 1)] for k in range(3)]
    e2 = [[_E(lmn3[k], lmn4[k], t, C[k] - D[k], c, d) for t in range(lmn3[k] + lmn4[k] + 1)] for k in range(3)]
    val = 0.0
    for t, ext in enumerate(e1[0]):
        for u, eyu in enumerate(e1[1]):
            for v, ezv in enumerate(e1[2]):
                for tau, fx in enumerate(e2[0]):
                    for nu, fy in enumerate(e2[1]):
                        for phi, fz in enumerate(e2[2]):
                            val += (ext * eyu * ezv * fx * fy * fz * (-1) ** (tau + nu + phi)
                                    * _R(t + tau, u + nu, v + phi, 0, alpha, pq[0], pq[1], pq[2], rpq2))
    return val * 2 * math.pi**2.5 / (p * q * math.sqrt(p + q))


def _contract(f1, f2, prim, *extra):
    A, l1, e1, c1 = f1
    B, l2, e2, c2 = f2
    s = 0.0
    for a, ca in zip(e1, c1):
        for b, cb in zip(e2, c2):
            s += ca * cb * prim(a, l1, A, b, l2, B, *extra)
    return s


def one_electron(basis: Basis):
    """(S, T, V_nuc) matrices."""
    n = basis.nao
    S, T, V = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1):
            fi, fj = basis.funcs[i], basis.funcs[j]
            S[i, j] = S[j, i] = _contract(fi, fj, _overlap_prim)
            T[i, j] = T[j, i] = _contract(fi, fj, _kinetic_prim)
            v = 0.0
            for sym, pos in basis.atoms:
                v -= CHARGE[sym] * _contract(fi, fj, _nuclear_prim, pos)
            V[i, j] = V[j, i] = v
    return S, T, V


def two_electron(basis: Basis) -> np.ndarray:
    """Dense chemist-notation (pq|rs), filled from the 8-fold unique quartets."""
    n = basis.nao
    eri = np.zeros((n, n, n, n))
    fs = basis.funcs
    for p in range(n):
        for q in range(p + 1):
            pq = p * (p + 1) // 2 + q
            for r in range(n):
                for s in range(r + 1):
                    if r * (r + 1) // 2 + s > pq:
                        continue
                    val = 0.0
                    A, l1, e1, c1 = fs[p]
                    B, l2, e2, c2 = fs[q]
                    C, l3, e3, c3 = fs[r]
                    D, l4, e4, c4 = fs[s]
                    for a, ca in zip(e1, c1):
                        for b, cb in zip(e2, c2):
                            for c, cc in zip(e3, c3):
                                for d, cd in zip(e4, c4):
                                    val += ca * cb * cc * cd * _eri_prim(a, l1, A, b, l2, B, c, l3, C, d, l4, D)
                    for (i, j, k, l) in ((p, q, r, s), (q, p, r, s), (p, q, s, r), (q, p, s, r),
                                         (r, s, p, q), (s, r, p, q), (r, s, q, p), (s, r, q, p)):
                        eri[i, j, k, l] = val
    return eri


def nuclear_repulsion(atoms) -> float:
    e = 0.0
    for i in range(len(atoms)):
        for j in range(i):
            e += CHARGE[atoms[i][0]] * CHARGE[atoms[j][0]] / np.linalg.norm(atoms[i][1] - atoms[j][1])
    return e


# ------------------------------------------------------------------------------------------------
# general angular momentum: AOs as weighted sums of Cartesian monomials over one radial contraction
_SOLID = {
    0: [[(1.0, (0, 0, 0))]],
    1: [[(1.0, (1, 0, 0))], [(1.0, (0, 1, 0))], [(1.0, (0, 0, 1))]],
    2: [  # r^2 Y_2m without the common sqrt(1/4pi) -- restored by _angular_norm
        [(math.sqrt(15.0), (1, 1, 0))],
        [(math.sqrt(15.0), (0, 1, 1))],
        [(math.sqrt(5.0) / 2.0 * 2.0, (0, 0, 2)), (-math.sqrt(5.0) / 2.0, (2, 0, 0)), (-math.sqrt(5.0) / 2.0, (0, 2, 0))],
        [(math.sqrt(15.0), (1, 0, 1))],
        [(math.sqrt(15.0) / 2.0, (2, 0, 0)), (-math.sqrt(15.0) / 2.0, (0, 2, 0))],
    ],
    3: [  # r^3 Y_3m sqrt(4 pi):  y(3xx-yy), xyz, y(4zz-xx-yy), z(2zz-3xx-3yy), x(4zz-xx-yy), z(xx-yy), x(xx-3yy)
        [(3.0 * math.sqrt(35.0 / 8.0), (2, 1, 0)), (-math.sqrt(35.0 / 8.0), (0, 3, 0))],
        [(math.sqrt(105.0), (1, 1, 1))],
        [(4.0 * math.sqrt(21.0 / 8.0), (0, 1, 2)), (-math.sqrt(21.0 / 8.0), (2, 1, 0)), (-math.sqrt(21.0 / 8.0), (0, 3, 0))],
        [(2.0 * math.sqrt(7.0 / 4.0), (0, 0, 3)), (-3.0 * math.sqrt(7.0 / 4.0), (2, 0, 1)), (-3.0 * math.sqrt(7.0 / 4.0), (0, 2, 1))],
        [(4.0 * math.sqrt(21.0 / 8.0), (1, 0, 2)), (-math.sqrt(21.0 / 8.0), (3, 0, 0)), (-math.sqrt(21.0 / 8.0), (1, 2, 0))],
        [(math.sqrt(105.0 / 4.0), (2, 0, 1)), (-math.sqrt(105.0 / 4.0), (0, 2, 1))],
        [(math.sqrt(35.0 / 8.0), (3, 0, 0)), (-3.0 * math.sqrt(35.0 / 8.0), (1, 2, 0))],
    ],
}
_ANG_COMMON = {0: 1.0, 1: math.sqrt(3.0), 2: 1.0, 3: 1.0}  # l = 1: sqrt(3/4pi) x;  l >= 2 factors sit in _SOLID


class SphericalBasis:
    """AOs of a molecule from a table {symbol: [(l, exponents, coefficients), ...]}, l <= 3, real
    spherical d and f functions.  ``aos`` holds (centre, [(weight, lmn), ...], exponents, coefficients)."""

    def __init__(self, atoms, table):
        self.atoms = atoms
        self.aos = []
        self.ao_slices = []
        for iat, (sym, pos) in enumerate(atoms):
            start = len(self.aos)
            for ang, exps, coefs in table[sym]:
                exps = np.asarray(exps, dtype=float)
                c = np.asarray(coefs, dtype=float) * self._radial_norm(ang, exps)
                ovl = 0.0  # radial self-overlap of the contraction: Gamma(l + 3/2) / (2 (a + b)^(l + 3/2))
                for ci, ai in zip(c, exps):
                    for cj, aj in zip(c, exps):
                        ovl += ci * cj * math.gamma(ang + 1.5) / (2.0 * (ai + aj) ** (ang + 1.5))
                c = c / math.sqrt(ovl)
                ang_norm = _ANG_COMMON[ang] / math.sqrt(4.0 * math.pi)
                for terms in _SOLID[ang]:
                    self.aos.append((pos, [(w * ang_norm, lmn) for w, lmn in terms], exps, c))
            self.ao_slices.append([iat, iat + 1, start, len(self.aos)])
        self.nao = len(self.aos)

    @staticmethod
    def _radial_norm(ang, exps):
        # int_0^inf r^(2l) e^(-2 a r^2) r^2 dr = Gamma(l + 3/2) / (2 (2a)^(l + 3/2))
        return np.sqrt(2.0 * (2.0 * exps) ** (ang + 1.5) / math.gamma(ang + 1.5))


def _ao_pair(f1, f2, prim, *extra):
    A, t1, e1, c1 = f1
    B, t2, e2, c2 = f2
    s = 0.0
    for w1, l1 in t1:
        for w2, l2 in t2:
            s += w1 * w2 * _contract((A, l1, e1, c1), (B, l2, e2, c2), prim, *extra)
    return s


def one_electron_general(basis: SphericalBasis):
    n = basis.nao
    S, T, V = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1):
            fi, fj = basis.aos[i], basis.aos[j]
            S[i, j] = S[j, i] = _ao_pair(fi, fj, _overlap_prim)
            T[i, j] = T[j, i] = _ao_pair(fi, fj, _kinetic_prim)
            v = 0.0
            for sym, pos in basis.atoms:
                v -= CHARGE_ALL[sym] * _ao_pair(fi, fj, _nuclear_prim, pos)
            V[i, j] = V[j, i] = v
    return S, T, V


def overlap_cross_general(basis_a: SphericalBasis, basis_b: SphericalBasis):
    return np.array([[_ao_pair(fa, fb, _overlap_prim) for fb in basis_b.aos] for fa in basis_a.aos])


def eri_element_general(basis: SphericalBasis, p: int, q: int, r: int, s: int) -> float:
    """One (pq|rs): every monomial and primitive combination summed."""
    (A, t1, e1, c1), (B, t2, e2, c2), (C, t3, e3, c3), (D, t4, e4, c4) = (basis.aos[i] for i in (p, q, r, s))
    val = 0.0
    for w1, l1 in t1:
        for w2, l2 in t2:
            for w3, l3 in t3:
                for w4, l4 in t4:
                    acc = 0.0
                    for a, ca in zip(e1, c1):
                        for b, cb in zip(e2, c2):
                            for c, cc in zip(e3, c3):
                                for d, cd in zip(e4, c4):
                                    acc += ca * cb * cc * cd * _eri_prim(a, l1, A, b, l2, B, c, l3, C, d, l4, D)
                    val += w1 * w2 * w3 * w4 * acc
    return val


CHARGE_ALL = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}
