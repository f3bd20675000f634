"""Exchange-correlation quadrature for the embedding-potential producer.

The reference gets E_xc and v_xc from PySCF's ``dft.UKS.get_veff`` (numint + libxc) at
nbed/driver.py:155-191 (global B3LYP Kohn-Sham), :315-431 (``_subsystem_dft``: E_xc of the active,
environment and total densities), :845-852 (the embedding potential) and :1138-1231 (DFT-in-DFT).
They are INPUTS of the embedded-SCF hot path -- produced a handful of times per molecule -- so this
module is host code: atom-centred quadrature grids, AO values on them, and the functionals PySCF
would hand to libxc.  The Coulomb and exact-exchange parts of the Kohn-Sham matrix stay on the GPU
(``GpuUKS``: libnbx J/K); an ``XCProvider`` supplies the semi-local remainder
``(E_xc, v_xc)`` through ``GpuUKS(xc_provider=...)``.

Functionals (libxc definitions, spin-polarised):
  * ``lda``    Slater exchange + VWN(RPA) correlation                      ("lda,vwn_rpa")
  * ``lda,vwn`` Slater exchange + VWN5 correlation (libxc LDA_C_VWN: PySCF's default functional)
  * ``b3lyp``  0.08 Slater + 0.72 B88 + 0.19 VWN(RPA) + 0.81 LYP + 0.20 HF  (libxc XC_HYB_GGA_XC_B3LYP,
               what PySCF >= 2.3 means by "b3lyp": the warning captured in
               docs/source/notebooks/localization.ipynb cell 13)
  * ``hf``     no semi-local part, 100 % exact exchange
Energy densities are written once, in torch float64; their derivatives with respect to
(rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb) come from autograd, not from hand-derived formulas.

Grid: Becke fuzzy cells (three smoothing iterations, Bragg-Slater size adjustment) over
Gauss-Chebyshev radial shells (Becke's mapping r = R (1 + x) / (1 - x)) times a product angular rule
(Gauss-Legendre in cos(theta) x uniform in phi, exact to degree 2 n_theta - 1).  It is NOT PySCF's
grid (Treutler-Ahlrichs radial + pruned Lebedev angular): the two quadratures converge to the same
integrals, so energies agree to the residual error of PySCF's default grid level 3 (~1e-6..1e-5 Ha),
not bit for bit -- see DESIGN.md section 6.
"""

from __future__ import annotations

import math

import numpy as np

from . import integrals


def _torch():
    import torch

    return torch

#: Bragg-Slater radii (Angstrom) used for the radial scale and the cell-size adjustment (Becke 1988)
BRAGG = {"H": 0.35, "C": 0.70, "N": 0.65, "O": 0.60, "F": 0.50}

HYBRID_FRACTION = {"hf": 1.0, "b3lyp": 0.2, "lda": 0.0, "lda,vwn_rpa": 0.0, "lda,vwn": 0.0, "lda,vwn5": 0.0, "svwn": 0.0,
                   "slater": 0.0}


def hybrid_fraction(xc: str) -> float:
    key = str(xc).lower().replace(" ", "")
    if key not in HYBRID_FRACTION:
        raise ValueError(f"functional {xc!r} is not in nbed_amd.xc ({sorted(HYBRID_FRACTION)})")
    return HYBRID_FRACTION[key]


# ---------------------------------------------------------------------------------------------- grid
def _angular_rule(n_theta: int):
    """Unit vectors and weights (sum 4 pi) of the product rule: Gauss-Legendre in cos(theta), 2 n_theta
    equidistant azimuths (half a step off the axes so that no point sits on a nucleus-nucleus line)."""
    x, w = np.polynomial.legendre.leggauss(n_theta)
    n_phi = 2 * n_theta
    phi = (np.arange(n_phi) + 0.5) * (2.0 * math.pi / n_phi)
    st = np.sqrt(1.0 - x * x)
    pts = np.stack([np.outer(st, np.cos(phi)), np.outer(st, np.sin(phi)), np.outer(x, np.ones(n_phi))], axis=-1)
    wts = np.outer(w, np.full(n_phi, 2.0 * math.pi / n_phi))
    return pts.reshape(-1, 3), wts.reshape(-1)


def _radial_rule(n_rad: int, scale: float):
    """Gauss-Chebyshev (second kind) nodes mapped to (0, inf) by r = scale (1 + x) / (1 - x); weights
    include r^2 dr."""
    i = np.arange(1, n_rad + 1)
    x = np.cos(i * math.pi / (n_rad + 1))
    w = math.pi / (n_rad + 1) * np.sin(i * math.pi / (n_rad + 1)) ** 2 / np.sqrt(1.0 - x * x)
    r = scale * (1.0 + x) / (1.0 - x)
    dr = 2.0 * scale / (1.0 - x) ** 2
    return r, w * dr * r * r


def _hip_backend(device):
    """libnbx on a HIP device (None for the host: the torch expressions below are then the whole path)."""
    if _torch().device(device).type != "cuda":
        return None
    from .backend import get_backend

    return get_backend()


# Treutler-Ahlrichs / Lebedev scheme (``scheme="lebedev"``): the construction PySCF's ``dft.gen_grid`` documents as
# its default -- Treutler & Ahlrichs' M4 radial map (J. Chem. Phys. 102, 346 (1995)) of Gauss-Chebyshev nodes,
# Lebedev-Laikov angular rules pruned near the nucleus as NWChem does, Becke cells with Treutler's
# sqrt-of-Bragg-radius size adjustment -- written from the published formulas (scipy provides the Lebedev rules).
# Level 3: 50 radial shells for H, 75 for Li-Ne, 302 angular points.
_TA_XI = {"H": 0.8, "C": 1.1, "N": 0.9, "O": 0.9, "F": 0.9}
_TA_RAD = {0: (10, 15), 1: (30, 40), 2: (40, 60), 3: (50, 75), 4: (60, 90), 5: (70, 105), 6: (80, 120), 7: (90, 135),
           8: (100, 150), 9: (200, 200)}           # radial shells: (period 1, period 2)
_TA_ANG = {0: (11, 15), 1: (17, 23), 2: (23, 29), 3: (29, 29), 4: (35, 41), 5: (41, 47), 6: (47, 53), 7: (53, 59),
           8: (59, 59), 9: (65, 65)}               # Lebedev degree: (period 1, period 2)
_LEBEDEV_DEGREES = (3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 35, 41, 47, 53, 59, 65)


def _lebedev(degree: int):
    """(unit vectors (n, 3), weights summing to 4 pi) of the Lebedev rule of the given degree."""
    try:
        from scipy.integrate import lebedev_rule
    except ImportError as exc:  # SciPy < 1.15
        raise ImportError("the 'lebedev' grid scheme needs scipy.integrate.lebedev_rule (SciPy >= 1.15); pass "
                          "xc_grid={'scheme': 'product', ...} (n_rad, n_theta) to use the product grid instead") from exc

    x, w = lebedev_rule(int(degree))
    return np.ascontiguousarray(x.T), np.asarray(w)


def _ta_radial(n: int, xi: float):
    """Treutler-Ahlrichs M4: x_i = cos(i pi / (n + 1)), r = -xi / ln 2 (1 + x)^0.6 ln((1 - x) / 2); returns
    (r ascending, r^2 dr weights)."""
    step = math.pi / (n + 1)
    i = np.arange(1, n + 1)
    x = np.cos(i * step)
    a = xi / math.log(2.0) * (1.0 + x) ** 0.6
    r = -a * np.log((1.0 - x) / 2.0)
    dr = step * np.sin(i * step) * a * (-0.6 / (1.0 + x) * np.log((1.0 - x) / 2.0) + 1.0 / (1.0 - x))
    return r[::-1], (r * r * dr)[::-1]


def _nwchem_pruned_degrees(sym: str, r: np.ndarray, degree: int):
    """Lebedev degree per radial shell: the full rule between 0.9 (0.5 for H) and 3.5 (4.5) Bragg radii, smaller
    ones nearer the nucleus and one step smaller outside (NWChem's pruning scheme)."""
    idx = _LEBEDEV_DEGREES.index(degree)
    if idx < _LEBEDEV_DEGREES.index(11):
        return np.full(r.shape, degree)
    if idx == _LEBEDEV_DEGREES.index(11):
        ladder = (7, 9, 9, 9, 7)  # 26, 38, 38, 38, 26 points
    else:
        ladder = (11, 15, _LEBEDEV_DEGREES[idx - 1], degree, _LEBEDEV_DEGREES[idx - 1])
    alphas = (0.25, 0.5, 1.0, 4.5) if sym in ("H", "He") else (0.1667, 0.5, 0.9, 3.5)
    place = (r[:, None] / (BRAGG[sym] / integrals.BOHR) > np.asarray(alphas)[None, :]).sum(axis=1)
    return np.asarray(ladder)[place]


def _atomic_shells(atoms, scheme: str, n_rad: int, n_theta: int, level: int):
    """Per atom: (points relative to the nucleus (g, 3), weights (g,)) before the cell partition."""
    out = []
    if scheme == "lebedev":
        for sym, _ in atoms:
            period = 0 if sym in ("H", "He") else 1
            r, wr = _ta_radial(_TA_RAD[level][period], _TA_XI[sym])
            degs = _nwchem_pruned_degrees(sym, r, _TA_ANG[level][period])
            pts, wts = [], []
            for d in sorted(set(int(v) for v in degs)):
                ang, aw = _lebedev(d)
                sel = np.flatnonzero(degs == d)
                pts.append((r[sel, None, None] * ang[None, :, :]).reshape(-1, 3))
                wts.append((wr[sel, None] * aw[None, :]).reshape(-1))
            out.append((np.concatenate(pts), np.concatenate(wts)))
        return out
    ang_pts, ang_w = _angular_rule(n_theta)
    for sym, _ in atoms:
        rad = BRAGG[sym] / integrals.BOHR
        r, wr = _radial_rule(n_rad if sym != "H" else max(n_rad * 3 // 4, 24), rad if sym != "H" else 2 * rad)
        out.append(((r[:, None, None] * ang_pts[None, :, :]).reshape(-1, 3), (wr[:, None] * ang_w[None, :]).reshape(-1)))
    return out


def build_grid(atoms, n_rad: int | None = None, n_theta: int | None = None, device="cpu", scheme: str | None = None,
               level: int = 3):
    """(points (G, 3) in Bohr, weights (G,)) of the molecular grid.  The Becke cell functions -- a
    (points x atoms x atoms) product -- are evaluated on ``device``: by nbx_becke_share (one thread per point)
    on a GPU, with torch in blocks of points on the host.

    ``scheme``: "lebedev" (Treutler-Ahlrichs radial x pruned Lebedev angular, ``level`` 0-9: the construction PySCF
    documents for its default grid, level 3 = its default -- and the default here, so that grid-sensitive numbers
    come out as the reference's do: DESIGN.md section 6) or "product" (Becke radial map x Gauss-Legendre/uniform
    angular product rule with ``n_rad`` (96) shells per heavy atom and ``n_theta`` (28) polar angles; chosen
    implicitly when either is given)."""
    if scheme is None:
        scheme = "lebedev" if (n_rad is None and n_theta is None) else "product"
    n_rad = 96 if n_rad is None else int(n_rad)
    n_theta = 28 if n_theta is None else int(n_theta)
    t = _torch()
    dev = t.device(device)
    be = _hip_backend(dev)
    if scheme not in ("product", "lebedev"):
        raise ValueError(f"unknown grid scheme {scheme!r}")
    centres = np.array([pos for _, pos in atoms])
    radii = np.array([BRAGG[sym] / integrals.BOHR for sym, _ in atoms])
    natm = len(atoms)
    dist = np.linalg.norm(centres[:, None, :] - centres[None, :, :], axis=-1)
    if scheme == "lebedev":
        # Treutler's adjustment: Becke's a_ij = u / (u^2 - 1), u = (chi - 1) / (chi + 1), with chi = sqrt(R_i / R_j)
        # instead of R_i / R_j, i.e. a_ij = (sqrt(R_j / R_i) - sqrt(R_i / R_j)) / 4, |a| capped at 1/2
        sq = np.sqrt(radii)
        aij = np.clip(0.25 * (sq[None, :] / sq[:, None] - sq[:, None] / sq[None, :]), -0.5, 0.5)
    else:
        # Becke's size adjustment a_ij from the ratio of the Bragg radii (|a| capped at 1/2)
        chi = radii[:, None] / radii[None, :]
        uab = (chi - 1.0) / (chi + 1.0)
        aij = np.clip(uab / (uab * uab - 1.0), -0.5, 0.5)
    centres_d = t.as_tensor(centres).to(dev)
    aij_d = t.as_tensor(aij).to(dev)
    inv_dist = t.as_tensor(1.0 / (dist + np.eye(natm))).to(dev)
    off_diag = (1.0 - t.eye(natm, dtype=t.float64, device=dev))
    pts_all, w_all = [], []
    shells = _atomic_shells(atoms, scheme, n_rad, n_theta, level)
    for ia, (sym, pos) in enumerate(atoms):
        pts = pos[None, :] + shells[ia][0]
        w = shells[ia][1]
        if natm > 1 and be is not None:
            share = be.to_host(be.becke_share(be.asarray(pts), centres_d, aij_d, inv_dist, ia))
            w = w * share
        elif natm > 1:
            share = np.empty(pts.shape[0])
            for g0 in range(0, pts.shape[0], 1 << 15):
                p = t.as_tensor(pts[g0:g0 + (1 << 15)]).to(dev)
                rg = (p[:, None, :] - centres_d[None, :, :]).norm(dim=-1)              # (g, natm)
                mu = (rg[:, :, None] - rg[:, None, :]) * inv_dist[None]                # (g, i, j)
                f = mu + aij_d[None] * (1.0 - mu * mu)
                for _ in range(3):
                    f = 1.5 * f - 0.5 * f**3
                s = 0.5 * (1.0 - f) * off_diag[None] + (1.0 - off_diag)[None]          # j = i contributes 1
                cell = s.prod(dim=2)                                                   # (g, natm)
                share[g0:g0 + p.shape[0]] = (cell[:, ia] / cell.sum(dim=1)).cpu().numpy()
            w = w * share
        keep = np.abs(w) > 1e-22  # (some Lebedev rules carry a negative weight: 74, 230, 266 points)
        pts_all.append(pts[keep])
        w_all.append(w[keep])
    return np.concatenate(pts_all), np.concatenate(w_all)


# ---------------------------------------------------------------------------------------------- AOs on the grid
def eval_ao(basis: "integrals.Basis", pts: np.ndarray, deriv: int = 1):
    """AO values (G, nao) and, with ``deriv`` = 1, their gradients (3, G, nao) for contracted Cartesian
    shells x^l y^m z^n sum_k c_k exp(-a_k r^2) in the AO order of ``integrals.Basis``."""
    npts = pts.shape[0]
    ao = np.zeros((npts, basis.nao_cart))
    dao = np.zeros((3, npts, basis.nao_cart)) if deriv else None
    for sh, ao0 in zip(basis.shells, basis.shell_ao0):
        d = pts - sh.centre[None, :]
        r2 = np.einsum("gx,gx->g", d, d)
        ex = np.exp(-np.outer(r2, sh.exps))  # (G, nprim)
        for ic, lmn in enumerate(sh.cart):
            rad = ex @ sh.coefs[ic]                       # sum_k c_k e^{-a r^2}
            drad = ex @ (sh.coefs[ic] * (-2.0 * sh.exps))  # d/d(r^2) * 2 -> multiplies x_i
            poly = np.ones(npts)
            for ax in range(3):
                if lmn[ax]:
                    poly = poly * d[:, ax] ** lmn[ax]
            ao[:, ao0 + ic] = poly * rad
            if deriv:
                for ax in range(3):
                    g = poly * drad * d[:, ax]
                    if lmn[ax]:
                        lower = np.ones(npts)
                        for bx in range(3):
                            e = lmn[bx] - (1 if bx == ax else 0)
                            if e:
                                lower = lower * d[:, bx] ** e
                        g = g + lmn[ax] * lower * rad
                    dao[ax, :, ao0 + ic] = g
    if not basis.pure_cartesian:  # spherical d functions: combinations of the Cartesian components
        ao = ao @ basis.cart2ao.T
        if deriv:
            dao = dao @ basis.cart2ao.T
    return ao, dao


def eval_ao_torch(basis: "integrals.Basis", pts, deriv: int = 1):
    """``eval_ao`` with torch tensors on the device of ``pts`` (G, 3): (ao (G, nao), dao (3, G, nao))."""
    t = _torch()
    npts = pts.shape[0]
    kw = dict(dtype=t.float64, device=pts.device)
    be = _hip_backend(pts.device)
    if be is not None:  # nbx_eval_ao: one thread per point over all shells (the shell loop below is the host form)
        table = getattr(basis, "_ao_table_device", None)
        if table is None:
            table = basis._ao_table_device = be.ao_table(basis)
        ao, dao = be.eval_ao(pts.contiguous(), table, deriv=bool(deriv))
        if not basis.pure_cartesian:  # Cartesian components -> the working (spherical) AOs: libnbx GEMMs
            u = getattr(basis, "_cart2ao_device", None)
            if u is None:
                u = basis._cart2ao_device = t.as_tensor(basis.cart2ao, **kw).contiguous()
            ao = be.gemm(ao, u, "N", "T")
            if deriv:
                dao = be.gemm(dao, u, "N", "T")
        return ao, dao
    ao = t.zeros((npts, basis.nao_cart), **kw)
    dao = t.zeros((3, npts, basis.nao_cart), **kw) if deriv else None
    for sh, ao0 in zip(basis.shells, basis.shell_ao0):
        d = pts - t.as_tensor(sh.centre, **kw)[None, :]
        r2 = (d * d).sum(dim=1)
        exps = t.as_tensor(sh.exps, **kw)
        ex = t.exp(-r2[:, None] * exps[None, :])
        for ic, lmn in enumerate(sh.cart):
            coef = t.as_tensor(sh.coefs[ic], **kw)
            rad = ex @ coef
            drad = ex @ (coef * (-2.0 * exps))

            def mono(e):
                out = t.ones(npts, **kw)
                for ax in range(3):
                    if e[ax]:
                        out = out * d[:, ax] ** e[ax]
                return out

            poly = mono(lmn)
            ao[:, ao0 + ic] = poly * rad
            if deriv:
                for ax in range(3):
                    g = poly * drad * d[:, ax]
                    if lmn[ax]:
                        low = list(lmn)
                        low[ax] -= 1
                        g = g + lmn[ax] * mono(low) * rad
                    dao[ax, :, ao0 + ic] = g
    if not basis.pure_cartesian:
        u = t.as_tensor(basis.cart2ao, **kw)
        ao = ao @ u.T
        if deriv:
            dao = dao @ u.T
    return ao, dao


# ---------------------------------------------------------------------------------------------- functionals

def _slater(t, ra, rb):
    cx = 1.5 * (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0)
    return -cx * (ra ** (4.0 / 3.0) + rb ** (4.0 / 3.0))


def _b88_correction(t, ra, rb, saa, sbb):
    """Becke's 1988 gradient correction to the Slater exchange (beta = 0.0042)."""
    beta = 0.0042
    out = 0.0
    for r, s in ((ra, saa), (rb, sbb)):
        r43 = r ** (4.0 / 3.0)
        x = t.sqrt(s) / r43
        out = out - beta * r43 * x * x / (1.0 + 6.0 * beta * x * t.asinh(x))
    return out


def _vwn_rpa(t, ra, rb):
    """Vosko-Wilk-Nusair correlation, the fit to the RPA data (libxc LDA_C_VWN_RPA; Gaussian's "VWN3")."""
    rho = ra + rb
    zeta = (ra - rb) / rho
    x = ((3.0 / (4.0 * math.pi)) / rho) ** (1.0 / 6.0)  # sqrt(rs)

    def ec(a, x0, b, c):
        q = math.sqrt(4.0 * c - b * b)
        xx = x * x + b * x + c
        xx0 = x0 * x0 + b * x0 + c
        at = t.atan(q / (2.0 * x + b))
        return a * (t.log(x * x / xx) + 2.0 * b / q * at
                    - b * x0 / xx0 * (t.log((x - x0) ** 2 / xx) + 2.0 * (b + 2.0 * x0) / q * at))

    ec_p = ec(0.0310907, -0.409286, 13.0720, 42.7198)
    ec_f = ec(0.01554535, -0.743294, 20.1231, 101.578)
    fz = ((1.0 + zeta) ** (4.0 / 3.0) + (1.0 - zeta) ** (4.0 / 3.0) - 2.0) / (2.0 ** (4.0 / 3.0) - 2.0)
    return rho * (ec_p + fz * (ec_f - ec_p))


def _vwn5(t, ra, rb):
    """Vosko-Wilk-Nusair correlation, the fit to the Ceperley-Alder data with the spin interpolation of the
    paper (libxc LDA_C_VWN, what PySCF means by "vwn" -- its default functional is "lda,vwn")."""
    rho = ra + rb
    zeta = (ra - rb) / rho
    x = ((3.0 / (4.0 * math.pi)) / rho) ** (1.0 / 6.0)  # sqrt(rs)

    def ec(a, x0, b, c):
        q = math.sqrt(4.0 * c - b * b)
        xx = x * x + b * x + c
        xx0 = x0 * x0 + b * x0 + c
        at = t.atan(q / (2.0 * x + b))
        return a * (t.log(x * x / xx) + 2.0 * b / q * at
                    - b * x0 / xx0 * (t.log((x - x0) ** 2 / xx) + 2.0 * (b + 2.0 * x0) / q * at))

    ec_p = ec(0.0310907, -0.10498, 3.72744, 12.9352)
    ec_f = ec(0.01554535, -0.32500, 7.06042, 18.0578)
    alpha = ec(-1.0 / (6.0 * math.pi**2), -0.0047584, 1.13107, 13.0045)  # spin stiffness
    fz = ((1.0 + zeta) ** (4.0 / 3.0) + (1.0 - zeta) ** (4.0 / 3.0) - 2.0) / (2.0 ** (4.0 / 3.0) - 2.0)
    fpp0 = 4.0 / (9.0 * (2.0 ** (1.0 / 3.0) - 1.0))
    z4 = zeta**4
    return rho * (ec_p + alpha * fz / fpp0 * (1.0 - z4) + (ec_f - ec_p) * fz * z4)


def _lyp(t, ra, rb, saa, sab, sbb):
    """Lee-Yang-Parr correlation in the gradient-only form of Miehlich, Savin, Stoll and Preuss (1989)."""
    a, b, c, d = 0.04918, 0.132, 0.2533, 0.349
    rho = ra + rb
    r13 = rho ** (-1.0 / 3.0)
    denom = 1.0 + d * r13
    omega = t.exp(-c * r13) / denom * rho ** (-11.0 / 3.0)
    delta = c * r13 + d * r13 / denom
    cf = 0.3 * (3.0 * math.pi**2) ** (2.0 / 3.0)
    stot = saa + 2.0 * sab + sbb
    t1 = -a * 4.0 / denom * ra * rb / rho
    brace = (ra * rb * (2.0 ** (11.0 / 3.0) * cf * (ra ** (8.0 / 3.0) + rb ** (8.0 / 3.0))
                        + (47.0 / 18.0 - 7.0 * delta / 18.0) * stot
                        - (2.5 - delta / 18.0) * (saa + sbb)
                        - (delta - 11.0) / 9.0 * (ra / rho * saa + rb / rho * sbb))
             - 2.0 / 3.0 * rho * rho * stot
             + (2.0 / 3.0 * rho * rho - ra * ra) * sbb
             + (2.0 / 3.0 * rho * rho - rb * rb) * saa)
    return t1 - a * b * omega * brace


def energy_density(xc: str, ra, rb, saa, sab, sbb):
    """Semi-local exchange-correlation energy per volume (torch tensors); None for ``hf``."""
    t = _torch()
    key = str(xc).lower().replace(" ", "")
    if key == "hf":
        return None
    if key in ("lda", "lda,vwn_rpa"):
        return _slater(t, ra, rb) + _vwn_rpa(t, ra, rb)
    if key in ("lda,vwn", "lda,vwn5", "svwn"):
        return _slater(t, ra, rb) + _vwn5(t, ra, rb)
    if key == "slater":
        return _slater(t, ra, rb)
    if key == "b3lyp":
        return (0.8 * _slater(t, ra, rb) + 0.72 * _b88_correction(t, ra, rb, saa, sbb)
                + 0.19 * _vwn_rpa(t, ra, rb) + 0.81 * _lyp(t, ra, rb, saa, sab, sbb))
    raise ValueError(f"functional {xc!r} is not in nbed_amd.xc")


class XCProvider:
    """``(E_xc, v_xc (2,N,N))`` of a two-spin density matrix: the callable ``GpuUKS`` takes as
    ``xc_provider``.  Built once per molecule (grid and AO values are kept)."""

    RHO_FLOOR = 1e-14
    BLOCK = 1 << 16  # grid points per matmul block (bounds the temporaries, not the stored AO values)
    SPLIT = 256      # rows per piece of the (nao x G) (G x nao) product: a batched GEMM summed over pieces

    def __init__(self, atoms, basis: "integrals.Basis", xc: str, n_rad: int | None = None, n_theta: int | None = None,
                 device=None, scheme: str | None = None, level: int = 3):
        """``device``: where the AO values live and the density / potential contractions run -- a torch
        device; None picks ``cuda`` when there is one (a 148-function molecule on the default grid is
        3.2 million points: 16 GB of AO values and gradients, 0.5 TFLOP per Kohn-Sham cycle)."""
        t = _torch()
        self.xc = str(xc).lower().replace(" ", "")
        self.hyb = hybrid_fraction(self.xc)
        if device is None:
            device = "cuda" if t.cuda.is_available() else "cpu"
        self.device = t.device(device)
        self.points, self.weights = build_grid(atoms, n_rad, n_theta, device=self.device, scheme=scheme, level=level)
        self.nelec_last = None
        self._blocks = []  # host form: (ao (g, nao), dao (3, g, nao)) per block of grid points
        self._ao = self._dao = None  # HIP form: the whole grid, (G, nao) and (3, G, nao)
        npts = self.points.shape[0]
        be = _hip_backend(self.device)
        if self.xc != "hf" and be is not None:
            # on a GPU the three passes of an evaluation are libnbx kernels over the whole stored arrays
            # (csrc/xc.hip); the arrays are filled block by block (bounds eval_ao's temporaries)
            nao = int(basis.nao)
            self._ao, self._dao = be.empty((npts, nao)), be.empty((3, npts, nao))
            for g0 in range(0, npts, self.BLOCK):
                ao, dao = eval_ao_torch(basis, t.as_tensor(self.points[g0:g0 + self.BLOCK]).to(self.device))
                self._ao[g0:g0 + ao.shape[0]].copy_(ao)
                self._dao[:, g0:g0 + ao.shape[0]].copy_(dao)
            self._w = t.as_tensor(self.weights).to(self.device)
            return
        pad = (-npts) % self.SPLIT  # zero rows: the long-K product of __call__ splits into equal pieces
        if self.xc != "hf":
            for g0 in range(0, npts, self.BLOCK):
                ao, dao = eval_ao_torch(basis, t.as_tensor(self.points[g0:g0 + self.BLOCK]).to(self.device))
                if g0 + self.BLOCK >= npts and pad:
                    ao = t.cat([ao, ao.new_zeros(pad, ao.shape[1])])
                    dao = t.cat([dao, dao.new_zeros(3, pad, dao.shape[2])], dim=1)
                self._blocks.append((ao.contiguous(), dao.contiguous()))
        self._w = t.cat([t.as_tensor(self.weights), t.zeros(pad, dtype=t.float64)]).to(self.device)

    @property
    def ao(self) -> np.ndarray:
        """AO values on the whole grid (G, nao), on the host (tests integrate the overlap with them)."""
        if self._ao is not None:
            return self._ao.cpu().numpy()
        return np.concatenate([b[0].cpu().numpy() for b in self._blocks])[: self.points.shape[0]]

    def _call_hip(self, be, dm: np.ndarray):
        """One evaluation on libnbx: nbx_xc_rho (densities and gradients, both spins, c = ao D on the matrix cores),
        nbx_xc_functional (analytic energy density and derivatives, weights folded in, E_xc reduced on the device),
        nbx_xc_vmat (the potential matrix with its ``half`` factor built on the fly).  Three kernels and two small
        reductions per evaluation; the host sees E_xc, the electron count and the (2, nao, nao) result."""
        from ._nbx import XC_CODES

        dmd = be.asarray(0.5 * (dm + dm.transpose(0, 2, 1)))
        rho, grad = be.xc_rho(self._ao, self._dao, dmd)
        vr, vec, sums = be.xc_functional(XC_CODES[self.xc], rho, grad, self._w, self.RHO_FLOOR)
        vxc = be.xc_vmat(self._ao, self._dao, vr, vec)
        sums_h = be.to_host(sums)
        self.nelec_last = float(sums_h[1])
        return float(sums_h[0]), be.to_host(vxc)

    def __call__(self, dm):
        t = _torch()
        dm = np.asarray(dm, dtype=np.float64)
        if self.xc == "hf":
            return 0.0, np.zeros_like(dm)
        if self._ao is not None:
            return self._call_hip(_hip_backend(self.device), dm)
        # host form (the CPU suite's checker; also the definition the HIP kernels are tested against)
        dmd = t.as_tensor(dm).to(self.device)
        dmd = 0.5 * (dmd + dmd.transpose(1, 2))
        rho, grad = [[], []], [[], []]
        for ao, dao in self._blocks:
            for x in range(2):
                c = ao @ dmd[x]                                            # (g, nao)
                rho[x].append((c * ao).sum(dim=1))
                grad[x].append(2.0 * (dao * c[None]).sum(dim=2))           # D symmetric
        rho = [t.cat(r) for r in rho]
        ga, gb = (t.cat(g, dim=1) for g in grad)
        w = self._w
        self.nelec_last = float((w * (rho[0] + rho[1])).sum())
        keep = ((rho[0] + rho[1]) > self.RHO_FLOOR).to(t.float64)          # drop the empty tail of the grid
        floor, tiny = self.RHO_FLOOR * 0.5, 1e-40
        ra = t.clamp(rho[0], min=floor).requires_grad_(True)
        rb = t.clamp(rho[1], min=floor).requires_grad_(True)
        saa = ((ga * ga).sum(dim=0) + tiny).requires_grad_(True)
        sab = (ga * gb).sum(dim=0).requires_grad_(True)
        sbb = ((gb * gb).sum(dim=0) + tiny).requires_grad_(True)
        exc = (w * keep * energy_density(self.xc, ra, rb, saa, sab, sbb)).sum()
        vra, vrb, vsaa, vsab, vsbb = (g if g is not None else t.zeros_like(w) for g in t.autograd.grad(
            exc, (ra, rb, saa, sab, sbb), allow_unused=True))
        # autograd differentiated sum_g w_g e_g: the derivatives already carry the weights
        vxc = t.zeros_like(dmd)
        for x, (vr, vs_same, gs, go) in enumerate(((vra, vsaa, ga, gb), (vrb, vsbb, gb, ga))):
            # v = v_rho phi_m phi_n + (2 v_ss grad rho_s + v_ab grad rho_other) . grad(phi_m phi_n)
            vec = (2.0 * vs_same * gs + vsab * go).contiguous()            # (3, G), weights included
            vr = vr.contiguous()
            g0 = 0
            for ao, dao in self._blocks:
                g1 = g0 + ao.shape[0]
                half = 0.5 * vr[g0:g1, None] * ao + (vec[:, g0:g1, None] * dao).sum(dim=0)
                n = ao.shape[1]  # K = grid points is long and M = N = nao short: split K, batch, sum
                vxc[x] += (ao.view(-1, self.SPLIT, n).transpose(1, 2) @ half.view(-1, self.SPLIT, n)).sum(dim=0)
                g0 = g1
        vxc = vxc + vxc.transpose(1, 2)
        return float(exc.detach()), vxc.cpu().numpy()
