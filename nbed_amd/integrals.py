"""Host-side Gaussian integrals for small s/p molecules.

The reference gets S, T, V_nuc and (pq|rs) from PySCF/libcint through ``gto.Mole.intor`` /
``get_ovlp`` / ``get_hcore`` (nbed/driver.py:86-104, nbed/localizers/occupied/spade.py:89-92,
nbed/localizers/virtual/concentric.py:83-88).  They are inputs of the hot path, produced once per
molecule; this module makes them without PySCF for contracted Cartesian s and p shells
(McMurchie-Davidson: Hermite expansion of the pair densities, Boys function), vectorised over the
primitive combinations of a shell block.  Enough for the reference's own CPU-runnable test case
(water / STO-3G, BASELINE configs[0]) and the other H/C/N/O/F molecules of its test set in STO-3G; larger bases
need the PySCF provider (``nbed_amd.driver.PySCFProvider``).

Conventions are PySCF's: Bohr radius 0.52917721092 Angstrom, AO order per atom = shells in basis
order, p functions as (x, y, z), normalised contracted functions, (pq|rs) in chemist order.
"""

from __future__ import annotations

import itertools
import math

import numpy as np
from scipy.special import hyp1f1

BOHR = 0.52917721092

_STO3G_1S = (0.15432897, 0.53532814, 0.44463454)
_STO3G_2S = (-0.09996723, 0.39951283, 0.70011547)
_STO3G_2P = (0.15591627, 0.60768372, 0.39195739)


def _sto3g_second_row(core, valence):
    return [(0, core, _STO3G_1S), (0, valence, _STO3G_2S), (1, valence, _STO3G_2P)]


#: basis name -> element -> [(l, exponents, contraction coefficients)]
#: STO-3G exponents are zeta^2 times the universal three-Gaussian fit of a Slater 1s / 2sp function
#: (tests/test_host_integrals.py checks the table against that identity)
BASIS_SETS = {
    "sto-3g": {
        "H": [(0, (3.42525091, 0.62391373, 0.16885540), _STO3G_1S)],
        "C": _sto3g_second_row((71.6168370, 13.0450960, 3.5305122), (2.9412494, 0.6834831, 0.2222899)),
        "N": _sto3g_second_row((99.1061690, 18.0523120, 4.8856602), (3.7804559, 0.8784966, 0.2857144)),
        "O": _sto3g_second_row((130.7093200, 23.8088610, 6.4436083), (5.0331513, 1.1695961, 0.3803890)),
        "F": _sto3g_second_row((166.6791300, 30.3608120, 8.2168207), (6.4648032, 1.5022812, 0.4885885)),
    }
}
NUCLEAR_CHARGE = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}
#: Slater exponents behind the table (1s, 2sp)
STO3G_ZETA = {"H": (1.24, None), "C": (5.67, 1.72), "N": (6.67, 1.95), "O": (7.66, 2.25), "F": (8.65, 2.55)}
_CART = {0: [(0, 0, 0)], 1: [(1, 0, 0), (0, 1, 0), (0, 0, 1)]}


def parse_geometry(xyz: str, unit: str = "angstrom"):
    """Raw xyz text ('<n>\\n<comment>\\n<sym> x y z' lines) -> [(symbol, position in Bohr)]."""
    scale = 1.0 if unit.lower().startswith(("b", "au")) else 1.0 / BOHR
    atoms = []
    for line in xyz.strip().splitlines()[2:]:
        if line.strip():
            sym, x, y, z = line.split()[:4]
            atoms.append((sym.capitalize(), np.array([float(x), float(y), float(z)]) * scale))
    return atoms


def supports(xyz: str, basis: str) -> bool:
    table = BASIS_SETS.get(basis.lower().replace("_", "-"))
    if table is None:
        return False
    try:
        return all(sym in table for sym, _ in parse_geometry(xyz))
    except ValueError:
        return False


def _dfact(n: int) -> int:
    return 1 if n <= 0 else n * _dfact(n - 2)


class Shell:
    """One contracted shell: centre, angular momentum, exponents and the coefficients of each of
    its Cartesian components including primitive and contracted normalisation."""

    def __init__(self, centre, ang, exps, coefs):
        self.centre = np.asarray(centre, dtype=float)
        self.ang = ang
        self.exps = np.asarray(exps, dtype=float)
        self.cart = _CART[ang]
        self.coefs = np.array([self._normalised(lmn, np.asarray(coefs, dtype=float)) for lmn in self.cart])

    def _normalised(self, lmn, coefs):
        big_l = sum(lmn)
        dd = math.prod(_dfact(2 * k - 1) for k in lmn)
        prim = np.sqrt(2.0 ** (2 * big_l + 1.5) * self.exps ** (big_l + 1.5) / (dd * math.pi ** 1.5))
        c = coefs * prim
        pair = c[:, None] * c[None, :] / (self.exps[:, None] + self.exps[None, :]) ** (big_l + 1.5)
        return c / math.sqrt(math.pi ** 1.5 * dd / 2.0 ** big_l * pair.sum())


class Basis:
    def __init__(self, atoms, basis: str = "sto-3g"):
        table = BASIS_SETS[basis.lower().replace("_", "-")]
        self.atoms = atoms
        self.shells: list[Shell] = []
        self.shell_ao0: list[int] = []
        self.ao_slices = []
        nao = nsh = 0
        for iat, (sym, pos) in enumerate(atoms):
            ao0, sh0 = nao, nsh
            for ang, exps, coefs in table[sym]:
                self.shells.append(Shell(pos, ang, exps, coefs))
                self.shell_ao0.append(nao)
                nao += len(_CART[ang])
                nsh += 1
            self.ao_slices.append([sh0, nsh, ao0, nao])
        self.nao = nao


def _hermite_e(imax, jmax, a, b, q):
    """E[i][j][t] (arrays over the primitive pairs) for one Cartesian direction: the expansion of
    x_A^i x_B^j exp(-a x_A^2 - b x_B^2) in Hermite Gaussians about P; q = A - B."""
    p = a + b
    xpa, xpb = -b / p * q, a / p * q
    e = {(0, 0, 0): np.exp(-a * b / p * q * q)}

    def get(i, j, t):
        return e.get((i, j, t), 0.0) if 0 <= t <= i + j else 0.0

    for i in range(imax + 1):
        for j in range(jmax + 1):
            if i == j == 0:
                continue
            for t in range(i + j + 1):
                if i > 0:  # raise i
                    e[(i, j, t)] = get(i - 1, j, t - 1) / (2 * p) + xpa * get(i - 1, j, t) + (t + 1) * get(i - 1, j, t + 1)
                else:  # raise j
                    e[(i, j, t)] = get(i, j - 1, t - 1) / (2 * p) + xpb * get(i, j - 1, t) + (t + 1) * get(i, j - 1, t + 1)
    return e


def _boys(n, x):
    return hyp1f1(n + 0.5, n + 1.5, -x) / (2.0 * n + 1.0)


def _hermite_r(tmax, alpha, rx, ry, rz):
    """R[(t,u,v)] = R^0_tuv for t + u + v <= tmax (arrays): Hermite Coulomb integrals."""
    r2 = rx * rx + ry * ry + rz * rz
    cur = {(0, 0, 0, n): (-2.0 * alpha) ** n * _boys(n, alpha * r2) for n in range(tmax + 1)}

    def rec(t, u, v, n):
        key = (t, u, v, n)
        if key in cur:
            return cur[key]
        if t > 0:
            val = rx * rec(t - 1, u, v, n + 1) + ((t - 1) * rec(t - 2, u, v, n + 1) if t > 1 else 0.0)
        elif u > 0:
            val = ry * rec(t, u - 1, v, n + 1) + ((u - 1) * rec(t, u - 2, v, n + 1) if u > 1 else 0.0)
        else:
            val = rz * rec(t, u, v - 1, n + 1) + ((v - 1) * rec(t, u, v - 2, n + 1) if v > 1 else 0.0)
        cur[key] = val
        return val

    return {(t, u, v): rec(t, u, v, 0) for t in range(tmax + 1) for u in range(tmax + 1 - t)
            for v in range(tmax + 1 - t - u)}


class _Pair:
    """Primitive-pair data of two shells (flattened over the exponent pairs)."""

    def __init__(self, sa: Shell, sb: Shell, extra: int = 0):
        a, b = np.meshgrid(sa.exps, sb.exps, indexing="ij")
        self.a, self.b = a.ravel(), b.ravel()
        self.p = self.a + self.b
        self.centre = (self.a[:, None] * sa.centre + self.b[:, None] * sb.centre) / self.p[:, None]
        q = sa.centre - sb.centre
        self.e = [_hermite_e(sa.ang, sb.ang + extra, self.a, self.b, q[x]) for x in range(3)]
        self.sa, self.sb = sa, sb

    def weights(self, ia, ib):
        """Contraction coefficient products of component ia of shell a and ib of shell b."""
        return (self.sa.coefs[ia][:, None] * self.sb.coefs[ib][None, :]).ravel()

    def hermite(self, lmn_a, lmn_b):
        """{(t,u,v): coefficient array} of the component pair."""
        out = {}
        for t in range(lmn_a[0] + lmn_b[0] + 1):
            for u in range(lmn_a[1] + lmn_b[1] + 1):
                for v in range(lmn_a[2] + lmn_b[2] + 1):
                    out[(t, u, v)] = (self.e[0][(lmn_a[0], lmn_b[0], t)] * self.e[1][(lmn_a[1], lmn_b[1], u)]
                                      * self.e[2][(lmn_a[2], lmn_b[2], v)])
        return out


def one_electron(basis: Basis):
    """(S, T, V_nuc), each (nao, nao)."""
    n = basis.nao
    s_mat, t_mat, v_mat = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
    for (ish, sa), (jsh, sb) in itertools.product(enumerate(basis.shells), repeat=2):
        pr = _Pair(sa, sb, extra=2)  # kinetic energy raises the ket by two
        pref = (math.pi / pr.p) ** 1.5
        # nuclear attraction: Hermite Coulomb integrals about every nucleus
        rs = []
        for sym, pos in basis.atoms:
            d = pr.centre - pos
            rs.append((NUCLEAR_CHARGE[sym], _hermite_r(sa.ang + sb.ang, pr.p, d[:, 0], d[:, 1], d[:, 2])))
        for ia, la in enumerate(sa.cart):
            for ib, lb in enumerate(sb.cart):
                w = pr.weights(ia, ib)

                def ovl(lb2):
                    if min(lb2) < 0:
                        return 0.0
                    return (pr.e[0][(la[0], lb2[0], 0)] * pr.e[1][(la[1], lb2[1], 0)] * pr.e[2][(la[2], lb2[2], 0)]) * pref

                s_val = ovl(lb)
                t_val = pr.b * (2 * sum(lb) + 3) * s_val
                for x in range(3):
                    up = tuple(lb[k] + 2 * (k == x) for k in range(3))
                    dn = tuple(lb[k] - 2 * (k == x) for k in range(3))
                    t_val = t_val - 2.0 * pr.b ** 2 * ovl(up) - 0.5 * lb[x] * (lb[x] - 1) * ovl(dn)
                v_val = 0.0
                herm = pr.hermite(la, lb)
                for charge, r in rs:
                    v_val = v_val - charge * sum(c * r[k] for k, c in herm.items())
                v_val = v_val * 2.0 * math.pi / pr.p
                i, j = basis.shell_ao0[ish] + ia, basis.shell_ao0[jsh] + ib
                s_mat[i, j] = np.dot(w, s_val)
                t_mat[i, j] = np.dot(w, t_val)
                v_mat[i, j] = np.dot(w, v_val)
    return s_mat, t_mat, v_mat


def two_electron(basis: Basis) -> np.ndarray:
    """(pq|rs), dense (nao,)*4 in chemist order; the 8-fold symmetry is used over shell quartets."""
    n = basis.nao
    eri = np.zeros((n, n, n, n))
    nsh = len(basis.shells)
    pairs = {(i, j): _Pair(basis.shells[i], basis.shells[j]) for i in range(nsh) for j in range(i + 1)}
    pair_list = sorted(pairs)
    for ip, (i, j) in enumerate(pair_list):
        bra = pairs[(i, j)]
        for k, l in pair_list[: ip + 1]:
            ket = pairs[(k, l)]
            ltot = bra.sa.ang + bra.sb.ang + ket.sa.ang + ket.sb.ang
            p, q = bra.p[:, None], ket.p[None, :]
            alpha = p * q / (p + q)
            d = bra.centre[:, None, :] - ket.centre[None, :, :]
            r = _hermite_r(ltot, alpha, d[..., 0], d[..., 1], d[..., 2])
            pref = 2.0 * math.pi ** 2.5 / (p * q * np.sqrt(p + q))
            for (ia, la), (ib, lb) in itertools.product(enumerate(bra.sa.cart), enumerate(bra.sb.cart)):
                hb = bra.hermite(la, lb)
                wb = bra.weights(ia, ib)
                for (ic, lc), (id_, ld) in itertools.product(enumerate(ket.sa.cart), enumerate(ket.sb.cart)):
                    hk = ket.hermite(lc, ld)
                    wk = ket.weights(ic, id_)
                    acc = 0.0
                    for (t, u, v), cb in hb.items():
                        for (t2, u2, v2), ck in hk.items():
                            sign = -1.0 if (t2 + u2 + v2) % 2 else 1.0
                            acc = acc + sign * (cb[:, None] * ck[None, :]) * r[(t + t2, u + u2, v + v2)]
                    val = float(wb @ (acc * pref) @ wk)
                    a0, b0 = basis.shell_ao0[i] + ia, basis.shell_ao0[j] + ib
                    c0, d0 = basis.shell_ao0[k] + ic, basis.shell_ao0[l] + id_
                    for (w, x), (y, z) in itertools.product(((a0, b0), (b0, a0)), ((c0, d0), (d0, c0))):
                        eri[w, x, y, z] = val
                        eri[y, z, w, x] = val
    return eri


def nuclear_repulsion(atoms) -> float:
    e = 0.0
    for (sa, ra), (sb, rb) in itertools.combinations(atoms, 2):
        e += NUCLEAR_CHARGE[sa] * NUCLEAR_CHARGE[sb] / float(np.linalg.norm(ra - rb))
    return e


def molecule_integrals(xyz: str, basis: str = "sto-3g", unit: str = "angstrom") -> dict:
    """Everything the embedding driver needs of a molecule: S, hcore = T + V, (pq|rs), e_nuc, the
    per-atom AO slices and the electron count of the neutral molecule."""
    atoms = parse_geometry(xyz, unit)
    bs = Basis(atoms, basis)
    s_mat, t_mat, v_mat = one_electron(bs)
    return {
        "S": s_mat, "T": t_mat, "V": v_mat, "hcore": t_mat + v_mat, "eri": two_electron(bs),
        "e_nuc": nuclear_repulsion(atoms), "ao_slices": bs.ao_slices, "nao": bs.nao,
        "nelectron": sum(NUCLEAR_CHARGE[s] for s, _ in atoms),
    }
