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
# 6-31G (Hehre, Ditchfield, Pople 1972): a six-primitive core s function and a 3 + 1 split valence with
# shared s/p exponents; cc-pVDZ (Dunning 1989) with one spherical d (p for hydrogen) polarisation shell.
_631G = {
    "H": [(0, (18.7311370, 2.8253937, 0.6401217), (0.03349460, 0.23472695, 0.81375733)),
          (0, (0.1612778,), (1.0,))],
    "C": [(0, (3047.5249, 457.36951, 103.94869, 29.210155, 9.2866630, 3.1639270),
           (0.0018347, 0.0140373, 0.0688426, 0.2321844, 0.4679413, 0.3623120)),
          (0, (7.8682724, 1.8812885, 0.5442493), (-0.1193324, -0.1608542, 1.1434564)),
          (1, (7.8682724, 1.8812885, 0.5442493), (0.0689991, 0.3164240, 0.7443083)),
          (0, (0.1687144,), (1.0,)), (1, (0.1687144,), (1.0,))],
    "N": [(0, (4173.5110, 627.45790, 142.90210, 40.234330, 12.820210, 4.3904370),
           (0.0018348, 0.0139950, 0.0685870, 0.2322410, 0.4690700, 0.3604550)),
          (0, (11.626358, 2.7162800, 0.7722180), (-0.1149610, -0.1691180, 1.1458520)),
          (1, (11.626358, 2.7162800, 0.7722180), (0.0675800, 0.3239070, 0.7408950)),
          (0, (0.2120313,), (1.0,)), (1, (0.2120313,), (1.0,))],
    "O": [(0, (5484.6717, 825.23495, 188.04696, 52.964500, 16.897570, 5.7996353),
           (0.0018311, 0.0139501, 0.0684451, 0.2327143, 0.4701930, 0.3585209)),
          (0, (15.539616, 3.5999336, 1.0137618), (-0.1107775, -0.1480263, 1.1307670)),
          (1, (15.539616, 3.5999336, 1.0137618), (0.0708743, 0.3397528, 0.7271586)),
          (0, (0.2700058,), (1.0,)), (1, (0.2700058,), (1.0,))],
}
_CCPVDZ = {
    "H": [(0, (13.01, 1.962, 0.4446, 0.1220), (0.019685, 0.137977, 0.478148, 0.501240)),
          (0, (0.1220,), (1.0,)), (1, (0.727,), (1.0,))],
    "C": [(0, (6665.0, 1000.0, 228.0, 64.71, 21.06, 7.495, 2.797, 0.5215, 0.1596),
           (0.000692, 0.005329, 0.027077, 0.101718, 0.274740, 0.448564, 0.285074, 0.015204, -0.003191)),
          (0, (6665.0, 1000.0, 228.0, 64.71, 21.06, 7.495, 2.797, 0.5215, 0.1596),
           (-0.000146, -0.001154, -0.005725, -0.023312, -0.063955, -0.149981, -0.127262, 0.544529, 0.580496)),
          (0, (0.1596,), (1.0,)),
          (1, (9.439, 2.002, 0.5456, 0.1517), (0.038109, 0.209480, 0.508557, 0.468842)),
          (1, (0.1517,), (1.0,)), (2, (0.550,), (1.0,))],
    "O": [(0, (11720.0, 1759.0, 400.8, 113.7, 37.03, 13.27, 5.025, 1.013, 0.3023),
           (0.000710, 0.005470, 0.027837, 0.104800, 0.283062, 0.448719, 0.270952, 0.015458, -0.002585)),
          (0, (11720.0, 1759.0, 400.8, 113.7, 37.03, 13.27, 5.025, 1.013, 0.3023),
           (-0.000160, -0.001263, -0.006267, -0.025716, -0.070924, -0.165411, -0.116955, 0.557368, 0.572759)),
          (0, (0.3023,), (1.0,)),
          (1, (17.70, 3.854, 1.046, 0.2753), (0.043018, 0.228913, 0.508728, 0.460531)),
          (1, (0.2753,), (1.0,)), (2, (1.185,), (1.0,))],
}
# 6-31G*: one d shell (exponent 0.8) on the heavy atoms; PySCF evaluates it with five spherical components
_631GS = {sym: (shells + [(2, (0.8,), (1.0,))] if sym != "H" else shells) for sym, shells in _631G.items()}
# cc-pVTZ (Dunning 1989): [4s3p2d1f] for C, N, O and [3s2p1d] for H
_CCPVTZ = {
    "H": [(0, (33.87, 5.095, 1.159, 0.3258, 0.1027), (0.006068, 0.045308, 0.202822, 0.503903, 0.383421)),
          (0, (0.3258,), (1.0,)), (0, (0.1027,), (1.0,)),
          (1, (1.407,), (1.0,)), (1, (0.388,), (1.0,)), (2, (1.057,), (1.0,))],
    "C": [(0, (8236.0, 1235.0, 280.8, 79.27, 25.59, 8.997, 3.319, 0.9059, 0.3643, 0.1285),
           (0.000531, 0.004108, 0.021087, 0.081853, 0.234817, 0.434401, 0.346129, 0.039378, -0.008983, 0.002385)),
          (0, (8236.0, 1235.0, 280.8, 79.27, 25.59, 8.997, 3.319, 0.9059, 0.3643, 0.1285),
           (-0.000113, -0.000878, -0.004540, -0.018133, -0.055760, -0.126895, -0.170352, 0.140382, 0.598684, 0.395389)),
          (0, (0.9059,), (1.0,)), (0, (0.1285,), (1.0,)),
          (1, (18.71, 4.133, 1.200, 0.3827, 0.1209), (0.014031, 0.086866, 0.290216, 0.501008, 0.343406)),
          (1, (0.3827,), (1.0,)), (1, (0.1209,), (1.0,)),
          (2, (1.097,), (1.0,)), (2, (0.318,), (1.0,)), (3, (0.761,), (1.0,))],
    "N": [(0, (11420.0, 1712.0, 389.3, 110.0, 35.57, 12.54, 4.644, 1.293, 0.5118, 0.1787),
           (0.000523, 0.004045, 0.020775, 0.080727, 0.233074, 0.433501, 0.347472, 0.041262, -0.008508, 0.002384)),
          (0, (11420.0, 1712.0, 389.3, 110.0, 35.57, 12.54, 4.644, 1.293, 0.5118, 0.1787),
           (-0.000115, -0.000895, -0.004624, -0.018528, -0.057339, -0.132076, -0.172510, 0.151814, 0.599944, 0.387462)),
          (0, (1.293,), (1.0,)), (0, (0.1787,), (1.0,)),
          (1, (26.63, 5.948, 1.742, 0.5550, 0.1725), (0.014670, 0.091764, 0.298683, 0.498487, 0.337023)),
          (1, (0.5550,), (1.0,)), (1, (0.1725,), (1.0,)),
          (2, (1.654,), (1.0,)), (2, (0.469,), (1.0,)), (3, (1.093,), (1.0,))],
    "O": [(0, (15330.0, 2299.0, 522.4, 147.3, 47.55, 16.76, 6.207, 1.752, 0.6882, 0.2384),
           (0.000508, 0.003929, 0.020243, 0.079181, 0.230687, 0.433118, 0.350260, 0.042728, -0.008154, 0.002381)),
          (0, (15330.0, 2299.0, 522.4, 147.3, 47.55, 16.76, 6.207, 1.752, 0.6882, 0.2384),
           (-0.000115, -0.000895, -0.004636, -0.018724, -0.058463, -0.136463, -0.175740, 0.160934, 0.603418, 0.378765)),
          (0, (1.752,), (1.0,)), (0, (0.2384,), (1.0,)),
          (1, (34.46, 7.749, 2.280, 0.7156, 0.2140), (0.015928, 0.099740, 0.310492, 0.491026, 0.336337)),
          (1, (0.7156,), (1.0,)), (1, (0.2140,), (1.0,)),
          (2, (2.314,), (1.0,)), (2, (0.645,), (1.0,)), (3, (1.428,), (1.0,))],
}
_CCPVDZ["N"] = [
    (0, (9046.0, 1357.0, 309.3, 87.73, 28.56, 10.21, 3.838, 0.7466, 0.2248),
     (0.000700, 0.005389, 0.027406, 0.103207, 0.278723, 0.448540, 0.278238, 0.015440, -0.002864)),
    (0, (9046.0, 1357.0, 309.3, 87.73, 28.56, 10.21, 3.838, 0.7466, 0.2248),
     (-0.000153, -0.001208, -0.005992, -0.024544, -0.067459, -0.158078, -0.121831, 0.549003, 0.578815)),
    (0, (0.2248,), (1.0,)),
    (1, (13.55, 2.917, 0.7973, 0.2185), (0.039919, 0.217169, 0.510319, 0.462214)),
    (1, (0.2185,), (1.0,)), (2, (0.817,), (1.0,))]
BASIS_SETS["cc-pvtz"] = _CCPVTZ
BASIS_SETS["ccpvtz"] = _CCPVTZ
BASIS_SETS["6-31g"] = _631G
BASIS_SETS["6-31g*"] = _631GS
BASIS_SETS["6-31g(d)"] = _631GS
BASIS_SETS["cc-pvdz"] = _CCPVDZ
BASIS_SETS["ccpvdz"] = _CCPVDZ
NUCLEAR_CHARGE = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}
#: Slater exponents behind the table (1s, 2sp)
STO3G_ZETA = {"H": (1.24, None), "C": (5.67, 1.72), "N": (6.67, 1.95), "O": (7.66, 2.25), "F": (8.65, 2.55)}
_CART = {0: [(0, 0, 0)], 1: [(1, 0, 0), (0, 1, 0), (0, 0, 1)],
         2: [(2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2)],
         3: [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3)]}
# real solid harmonics of l = 2 over (xx, xy, xz, yy, yz, zz), PySCF's order m = -2 .. 2:
# xy, yz, (2 zz - xx - yy) / 2, xz, (xx - yy) sqrt(3) / 2  (the sqrt(3) of the xy-type ones is absorbed
# by the numerical normalisation of each function)
_SPH = {2: np.array([[0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 1, 0], [-0.5, 0, 0, -0.5, 0, 1.0], [0, 0, 1, 0, 0, 0],
                     [1.0, 0, 0, -1.0, 0, 0]], dtype=float),
        # l = 3 over (xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz), m = -3 .. 3:
        # y(3xx - yy), xyz, y(4zz - xx - yy), z(2zz - 3xx - 3yy), x(4zz - xx - yy), z(xx - yy), x(xx - 3yy)
        3: np.array([[0, 3, 0, 0, 0, 0, -1, 0, 0, 0],
                     [0, 0, 0, 0, 1, 0, 0, 0, 0, 0],
                     [0, -1, 0, 0, 0, 0, -1, 0, 4, 0],
                     [0, 0, -3, 0, 0, 0, 0, -3, 0, 2],
                     [-1, 0, 0, -1, 0, 4, 0, 0, 0, 0],
                     [0, 0, 1, 0, 0, 0, 0, -1, 0, 0],
                     [1, 0, 0, -3, 0, 0, 0, 0, 0, 0]], dtype=float)}


def parse_geometry(xyz: str, unit: str = "angstrom"):
    """Raw xyz text ('<n>\\n<comment>\\n<sym> x y z' lines) -> [(symbol, position in Bohr)]."""
    scale = 1.0 if unit.lower().startswith(("b", "au")) else 1.0 / BOHR
    atoms = []
    for line in xyz.strip().splitlines()[2:]:
        if line.strip():
            sym, x, y, z = line.split()[:4]
            atoms.append((sym.capitalize(), np.array([float(x), float(y), float(z)]) * scale))
    return atoms


def supports(xyz: str, basis) -> bool:
    table = basis if isinstance(basis, dict) else BASIS_SETS.get(str(basis).lower().replace("_", "-"))
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

    def __init__(self, centre, ang, exps, coefs, cart: bool = False):
        self.centre = np.asarray(centre, dtype=float)
        self.ang = ang
        self.exps = np.asarray(exps, dtype=float)
        self.cart = _CART[ang]
        if ang <= 1:  # Cartesian = spherical: every component normalised in closed form
            self.coefs = np.array([self._normalised(lmn, np.asarray(coefs, dtype=float)) for lmn in self.cart])
            self.sph = np.eye(len(self.cart))
        else:
            # the contraction coefficients refer to normalised primitives: relative weight a^((2l+3)/4); the
            # spherical combinations are normalised numerically from the shell's own Cartesian overlap
            c = np.asarray(coefs, dtype=float) * self.exps ** ((2 * ang + 3) / 4.0)
            self.coefs = np.array([c for _ in self.cart])
            self.sph = np.eye(len(self.cart)) if cart else _SPH[ang].copy()
            ovl = _self_overlap(self)
            self.sph = self.sph / np.sqrt(np.einsum("mi,ij,mj->m", self.sph, ovl, self.sph))[:, None]

    def _normalised(self, lmn, coefs):
        big_l = sum(lmn)
        dd = math.prod(_dfact(2 * k - 1) for k in lmn)
        prim = np.sqrt(2.0 ** (2 * big_l + 1.5) * self.exps ** (big_l + 1.5) / (dd * math.pi ** 1.5))
        c = coefs * prim
        pair = c[:, None] * c[None, :] / (self.exps[:, None] + self.exps[None, :]) ** (big_l + 1.5)
        return c / math.sqrt(math.pi ** 1.5 * dd / 2.0 ** big_l * pair.sum())


def _self_overlap(sh):
    """Overlap of the Cartesian components of one shell with themselves (its own normalisation)."""
    a, b = np.meshgrid(sh.exps, sh.exps, indexing="ij")
    a, b = a.ravel(), b.ravel()
    p = a + b
    e = _hermite_e(sh.ang, sh.ang, a, b, 0.0)
    pref = (math.pi / p) ** 1.5
    n = len(sh.cart)
    out = np.zeros((n, n))
    for i, la in enumerate(sh.cart):
        for j, lb in enumerate(sh.cart):
            w = (sh.coefs[i][:, None] * sh.coefs[j][None, :]).ravel()
            out[i, j] = np.dot(w, e[(la[0], lb[0], 0)] * e[(la[1], lb[1], 0)] * e[(la[2], lb[2], 0)] * pref)
    return out


class Basis:
    """Shells of a molecule.  Integrals are evaluated over the CARTESIAN components (offsets
    ``shell_ao0``, ``nao_cart`` of them) and brought to the AOs proper -- spherical d functions, as
    PySCF's default -- by ``to_ao`` (the (nao, nao_cart) matrix ``cart2ao``; the identity for s/p bases)."""

    def __init__(self, atoms, basis="sto-3g", cart: bool = False):
        # a name of BASIS_SETS, or (as PySCF's mol.basis may be) a table {symbol: [(l, exponents, coefficients)]}
        table = basis if isinstance(basis, dict) else BASIS_SETS[basis.lower().replace("_", "-")]
        self.atoms = atoms
        self.shells: list[Shell] = []
        self.shell_ao0: list[int] = []
        self.ao_slices = []
        nao = ncart = nsh = 0
        blocks = []
        for iat, (sym, pos) in enumerate(atoms):
            ao0, sh0 = nao, nsh
            for ang, exps, coefs in table[sym]:
                sh = Shell(pos, ang, exps, coefs, cart)
                self.shells.append(sh)
                self.shell_ao0.append(ncart)
                blocks.append((nao, ncart, sh.sph))
                ncart += len(_CART[ang])
                nao += sh.sph.shape[0]
                nsh += 1
            self.ao_slices.append([sh0, nsh, ao0, nao])
        self.nao, self.nao_cart = nao, ncart
        self.cart2ao = np.zeros((nao, ncart))
        for a0, c0, t in blocks:
            self.cart2ao[a0:a0 + t.shape[0], c0:c0 + t.shape[1]] = t
        self.pure_cartesian = nao == ncart and bool(np.array_equal(self.cart2ao, np.eye(nao)))

    def to_ao(self, m: np.ndarray) -> np.ndarray:
        """Cartesian-component tensor (every axis of length nao_cart) -> AO tensor."""
        if self.pure_cartesian:
            return m
        u = self.cart2ao
        for ax in range(m.ndim):
            m = np.moveaxis(np.tensordot(u, m, axes=(1, ax)), 0, ax)
        return m


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
    n = basis.nao_cart
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
    return basis.to_ao(s_mat), basis.to_ao(t_mat), basis.to_ao(v_mat)


def overlap_cross(basis_a: Basis, basis_b: Basis) -> np.ndarray:
    """<a_i | b_j> between the AOs of two basis sets (possibly of different molecules / kinds):
    PySCF's ``gto.intor_cross('int1e_ovlp_sph', mol_a, mol_b)`` (nbed/localizers/virtual/concentric.py:83-88)."""
    out = np.zeros((basis_a.nao_cart, basis_b.nao_cart))
    for ish, sa in enumerate(basis_a.shells):
        for jsh, sb in enumerate(basis_b.shells):
            pr = _Pair(sa, sb)
            pref = (math.pi / pr.p) ** 1.5
            for ia, la in enumerate(sa.cart):
                for ib, lb in enumerate(sb.cart):
                    val = pr.e[0][(la[0], lb[0], 0)] * pr.e[1][(la[1], lb[1], 0)] * pr.e[2][(la[2], lb[2], 0)] * pref
                    out[basis_a.shell_ao0[ish] + ia, basis_b.shell_ao0[jsh] + ib] = np.dot(pr.weights(ia, ib), val)
    return basis_a.cart2ao @ out @ basis_b.cart2ao.T


def two_electron(basis: Basis) -> np.ndarray:
    """(pq|rs), dense (nao,)*4 in chemist order; the 8-fold symmetry is used over shell quartets."""
    n = basis.nao_cart
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
    return basis.to_ao(eri)


def nuclear_repulsion(atoms) -> float:
    e = 0.0
    for (sa, ra), (sb, rb) in itertools.combinations(atoms, 2):
        e += NUCLEAR_CHARGE[sa] * NUCLEAR_CHARGE[sb] / float(np.linalg.norm(ra - rb))
    return e


def _shell_arrays(basis: Basis):
    sh = basis.shells
    return (np.array([s.ang for s in sh], dtype=np.int32), np.array([len(s.exps) for s in sh], dtype=np.int32),
            np.array([s.sph.shape[0] for s in sh], dtype=np.int32),
            np.ascontiguousarray([s.centre for s in sh], dtype=np.float64),
            np.concatenate([s.exps for s in sh]).astype(np.float64),
            np.concatenate([s.coefs[0] for s in sh]).astype(np.float64),  # components of a shell share them
            np.concatenate([np.ascontiguousarray(s.sph, dtype=np.float64).ravel() for s in sh]))


def one_electron_native(basis: Basis, nthreads: int = 0):
    """(S, T, V_nuc) from libnbx's host engine (``nbx_host_1e``): ``one_electron`` in threaded C++."""
    import ctypes

    from . import _nbx

    lib = _nbx.load_library()
    ang, nprim, nfunc, centres, exps, coefs, sph = _shell_arrays(basis)
    charges = np.array([NUCLEAR_CHARGE[s] for s, _ in basis.atoms], dtype=np.float64)
    xyz = np.ascontiguousarray([p for _, p in basis.atoms], dtype=np.float64)
    out = [np.empty((basis.nao, basis.nao)) for _ in range(3)]
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _nbx.check(lib, lib.nbx_host_1e(len(basis.shells), ptr(ang), ptr(nprim), ptr(nfunc), ptr(centres), ptr(exps), ptr(coefs),
                                    ptr(sph), len(charges), ptr(charges), ptr(xyz), int(nthreads), *(ptr(o) for o in out)))
    return tuple(out)


def two_electron_native(basis: Basis, nthreads: int = 0, cutoff: float = 1e-16) -> np.ndarray:
    """(pq|rs) from libnbx's host engine (``nbx_host_eri``, csrc/ints_host.cpp: the same McMurchie-Davidson
    scheme in C++ over a thread pool) -- what makes a 148-function molecule practical.  Raises if libnbx.so
    is not built; ``two_electron`` above is the numpy engine it is tested against."""
    import ctypes

    from . import _nbx

    lib = _nbx.load_library()
    sh = basis.shells
    ang = np.array([s.ang for s in sh], dtype=np.int32)
    nprim = np.array([len(s.exps) for s in sh], dtype=np.int32)
    nfunc = np.array([s.sph.shape[0] for s in sh], dtype=np.int32)
    centres = np.ascontiguousarray([s.centre for s in sh], dtype=np.float64)
    exps = np.concatenate([s.exps for s in sh]).astype(np.float64)
    coefs = np.concatenate([s.coefs[0] for s in sh]).astype(np.float64)  # components of a shell share them
    sph = np.concatenate([np.ascontiguousarray(s.sph, dtype=np.float64).ravel() for s in sh])
    out = np.empty((basis.nao,) * 4)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _nbx.check(lib, lib.nbx_host_eri(len(sh), ptr(ang), ptr(nprim), ptr(nfunc), ptr(centres), ptr(exps), ptr(coefs),
                                     ptr(sph), float(cutoff), int(nthreads), ptr(out)))
    return out


def molecule_integrals(xyz: str, basis: str = "sto-3g", unit: str = "angstrom", cart: bool = False,
                       engine: str = "native") -> dict:
    """Everything the embedding driver needs of a molecule: S, hcore = T + V, (pq|rs), e_nuc, the
    per-atom AO slices and the electron count of the neutral molecule."""
    atoms = parse_geometry(xyz, unit)
    bs = Basis(atoms, basis, cart)  # cart: six Cartesian d functions (PySCF's mol.cart), default five spherical
    s_mat, t_mat, v_mat = one_electron_native(bs) if engine == "native" else one_electron(bs)
    return {
        "S": s_mat, "T": t_mat, "V": v_mat, "hcore": t_mat + v_mat,
        "eri": two_electron_native(bs) if engine == "native" else two_electron(bs),
        "e_nuc": nuclear_repulsion(atoms), "ao_slices": bs.ao_slices, "nao": bs.nao,
        "nelectron": sum(NUCLEAR_CHARGE[s] for s, _ in atoms),
    }
