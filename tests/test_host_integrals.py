"""The product's host-side integral engine (nbed_amd/integrals.py) against the oracle's
independent one (oracle/gto.py) and the committed water/STO-3G fixture -- whose SCF energy
reproduces the literal values of the reference's own test (tests/test_driver.py:52-61, see
test_water_kat.py) -- and the driver running end to end on the built-in provider (no PySCF, no
test provider)."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import gto as oracle_gto
from oracle_backend import OracleBackend

from nbed_amd import NbedConfig, integrals, nbed
from nbed_amd.driver import BuiltinHFProvider

WATER_XYZ = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"
H2O2_XYZ = ("4\n\nO   0.000  0.734  -0.052\nO   0.000  -0.734  -0.052\nH   0.839  0.881  0.419\n"
            "H   -0.839  -0.881  0.419")
H2_XYZ = "2\n\nH 0.0 0.0 0.0\nH 0.0 0.0 0.74"


def test_water_integrals_match_the_fixture():
    g = load_golden("water_sto3g")
    m = integrals.molecule_integrals(WATER_XYZ)
    assert m["nao"] == 7 and m["nelectron"] == 10
    np.testing.assert_allclose(m["S"], g["S"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(m["T"], g["T"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(m["V"], g["V"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(m["eri"], g["eri"], rtol=0, atol=1e-12)
    assert abs(m["e_nuc"] - float(g["ref_e_nuc"])) < 1e-12  # the reference's literal
    # AO ranges per atom (columns 2:4 of aoslice_by_atom; the shell columns here count shells as
    # PySCF does: O has three)
    assert [list(r[2:]) for r in m["ao_slices"]] == [list(map(int, r[2:])) for r in g["ao_slices"]]
    assert [list(r[:2]) for r in m["ao_slices"]] == [[0, 3], [3, 4], [4, 5]]


@pytest.mark.parametrize("xyz", [H2_XYZ, H2O2_XYZ])
def test_integrals_match_the_oracle_engine(xyz):
    m = integrals.molecule_integrals(xyz)
    atoms = oracle_gto.parse_xyz(xyz)
    basis = oracle_gto.Basis(atoms)
    s, t, v = oracle_gto.one_electron(basis)
    np.testing.assert_allclose(m["S"], s, rtol=0, atol=1e-13)
    np.testing.assert_allclose(m["T"], t, rtol=0, atol=1e-12)
    np.testing.assert_allclose(m["V"], v, rtol=0, atol=1e-11)
    eri = m["eri"]
    n = m["nao"]
    # permutational symmetry of the product's tensor, and a sample of elements against the oracle
    np.testing.assert_array_equal(eri, eri.transpose(1, 0, 2, 3))
    np.testing.assert_array_equal(eri, eri.transpose(2, 3, 0, 1))
    if n <= 4:
        np.testing.assert_allclose(eri, oracle_gto.two_electron(basis), rtol=0, atol=1e-12)
    assert abs(m["e_nuc"] - oracle_gto.nuclear_repulsion(atoms)) < 1e-12
    assert np.all(np.linalg.eigvalsh(s) > 0) and abs(np.diag(s) - 1).max() < 1e-12


def test_supports():
    assert integrals.supports(WATER_XYZ, "STO-3G") and integrals.supports(H2_XYZ, "sto-3g")
    assert integrals.supports(WATER_XYZ, "cc-pVDZ") and integrals.supports(H2O2_XYZ, "6-31G*")
    assert integrals.supports(WATER_XYZ, "cc-pVTZ")
    assert not integrals.supports(WATER_XYZ, "def2-SVP")
    assert not integrals.supports("2\n\nF 0 0 0\nH 0 0 0.9", "cc-pVDZ")  # no fluorine table in that set
    assert not integrals.supports("1\n\nS 0 0 0", "sto-3g")


def test_sto3g_table_is_zeta_scaled_universal_fit():
    """Every STO-3G exponent is zeta^2 times the universal fit of a Slater function: guards the
    table of the elements that have no reference literal (C, N, F) against typos."""
    fit_1s = np.array([2.227660584, 0.405771156, 0.109818])
    fit_2sp = np.array([0.994203, 0.231031, 0.0751386])
    for sym, (z1, z2) in integrals.STO3G_ZETA.items():
        shells = integrals.BASIS_SETS["sto-3g"][sym]
        np.testing.assert_allclose(shells[0][1], z1 * z1 * fit_1s, rtol=2e-5)
        if z2 is not None:
            assert shells[1][1] == shells[2][1] and shells[1][0] == 0 and shells[2][0] == 1
            np.testing.assert_allclose(shells[1][1], z2 * z2 * fit_2sp, rtol=2e-5)
    # methyl radical (the reference's open-shell test molecule): 8 basis functions, normalised
    m = integrals.molecule_integrals("4\n\nC 0 0 0\nH 1.079 0 0\nH -0.5395 0.9344 0\nH -0.5395 -0.9344 0")
    assert m["nao"] == 8 and m["nelectron"] == 9
    np.testing.assert_allclose(np.diag(m["S"]), 1.0, atol=1e-12)
    assert np.all(np.linalg.eigvalsh(m["S"]) > 0)


def test_d_shell_integrals_match_the_oracle_engine():
    """Water / 6-31G* (18 AOs, five spherical d functions on oxygen): the product's vectorised
    shell-pair code with numerically normalised spherical combinations against the oracle's scalar
    recursions with libcint's closed-form normalisation; and the cross-basis overlap the concentric
    localizer asks PySCF for (gto.intor_cross, nbed/localizers/virtual/concentric.py:83-88)."""
    m = integrals.molecule_integrals(WATER_XYZ, "6-31G*")
    atoms = oracle_gto.parse_xyz(WATER_XYZ)
    ob = oracle_gto.SphericalBasis(atoms, integrals.BASIS_SETS["6-31g*"])
    assert m["nao"] == ob.nao == 18 and [r[2:] for r in m["ao_slices"]] == [[0, 14], [14, 16], [16, 18]]
    s, t, v = oracle_gto.one_electron_general(ob)
    np.testing.assert_allclose(m["S"], s, rtol=0, atol=1e-13)
    np.testing.assert_allclose(m["T"], t, rtol=0, atol=1e-12)
    np.testing.assert_allclose(m["V"], v, rtol=0, atol=1e-11)
    eri = m["eri"]
    np.testing.assert_allclose(eri, eri.transpose(1, 0, 2, 3), rtol=0, atol=1e-14)
    np.testing.assert_allclose(eri, eri.transpose(2, 3, 0, 1), rtol=0, atol=1e-14)
    rng = np.random.default_rng(5)
    picks = [(9, 9, 9, 9), (11, 11, 13, 13), (9, 15, 12, 17), (11, 3, 11, 0), (13, 13, 14, 16), (10, 5, 10, 5)]
    picks += [tuple(int(x) for x in rng.integers(0, 18, 4)) for _ in range(40)]
    nonzero = 0
    for p, q, r, s_ in picks:
        ref = oracle_gto.eri_element_general(ob, p, q, r, s_)
        assert abs(ref - eri[p, q, r, s_]) < 1e-12
        nonzero += abs(ref) > 1e-6
    assert nonzero >= 15
    atoms_p = integrals.parse_geometry(WATER_XYZ)
    cross = integrals.overlap_cross(integrals.Basis(atoms_p, "6-31g"), integrals.Basis(atoms_p, "cc-pvdz"))
    ref = oracle_gto.overlap_cross_general(oracle_gto.SphericalBasis(atoms, integrals.BASIS_SETS["6-31g"]),
                                           oracle_gto.SphericalBasis(atoms, integrals.BASIS_SETS["cc-pvdz"]))
    assert cross.shape == (13, 24)
    np.testing.assert_allclose(cross, ref, rtol=0, atol=1e-13)
    same = integrals.Basis(atoms_p, "cc-pvdz")
    np.testing.assert_allclose(integrals.overlap_cross(same, same), integrals.one_electron(same)[0], rtol=0, atol=1e-14)


@pytest.mark.parametrize("xyz,basis,cart", [(WATER_XYZ, "6-31g*", False), (WATER_XYZ, "6-31g*", True),
                                            ("2\n\nH 0 0 0\nH 0 0 12.0", "cc-pvdz", False), (H2O2_XYZ, "sto-3g", False)])
def test_native_eri_engine_matches_the_numpy_engine(xyz, basis, cart):
    """libnbx's threaded host engine (nbx_host_eri, csrc/ints_host.cpp) against the shell-pair numpy code:
    spherical and Cartesian d shells, and a pair of atoms 12 A apart whose Coulomb integrals sit in the
    asymptotic branch of the Boys function (T ~ 1e3)."""
    bs = integrals.Basis(integrals.parse_geometry(xyz), basis, cart)
    ref = integrals.two_electron(bs)
    for nthreads in (1, 0):
        got = integrals.two_electron_native(bs, nthreads=nthreads)
        assert got.shape == ref.shape == (bs.nao,) * 4
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-13)
    np.testing.assert_array_equal(got, got.transpose(1, 0, 3, 2))
    np.testing.assert_array_equal(got, got.transpose(2, 3, 0, 1))
    # one-electron matrices of the same shells: nbx_host_1e against the numpy engine
    for a, b in zip(integrals.one_electron_native(bs, nthreads=2), integrals.one_electron(bs)):
        np.testing.assert_allclose(a, b, rtol=0, atol=5e-13)
        np.testing.assert_array_equal(a, a.T)


def test_native_eri_engine_rejects_bad_shells():
    import ctypes

    from nbed_amd import _nbx

    lib = _nbx.load_library()
    one = np.ones(1)
    i32 = lambda *v: np.array(v, dtype=np.int32)  # noqa: E731
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    out = np.zeros(1)
    args = lambda ang, nfunc: (1, ptr(i32(ang)), ptr(i32(1)), ptr(i32(nfunc)), ptr(np.zeros(3)), ptr(one), ptr(one),  # noqa: E731
                               ptr(np.ones(100)), 1e-16, 1, ptr(out))
    assert lib.nbx_host_eri(*args(4, 9)) == -1       # g shells are not covered
    assert lib.nbx_host_eri(*args(2, 4)) == -1       # a d shell has five or six functions
    assert lib.nbx_host_eri(*args(0, 1)) == 0 and out[0] > 0


def test_f_shell_integrals_match_the_oracle_engine():
    """p, d and f shells mixed on two centres (a hand-made table, two primitives in the p shell): one-
    electron matrices in full and sampled (pq|rs) of libnbx's engine against the oracle's closed-form
    real solid harmonics; the numpy engine on the f-only part."""
    table = {"C": [(1, (0.38, 0.9), (0.7, 0.4)), (2, (1.097,), (1.0,)), (3, (0.761,), (1.0,))],
             "H": [(0, (0.3,), (1.0,)), (2, (1.057,), (1.0,)), (3, (0.9,), (1.0,))]}
    xyz = "2\n\nC 0.1 -0.2 0.3\nH 0.9 0.5 -0.4"
    bs = integrals.Basis(integrals.parse_geometry(xyz), table)
    ob = oracle_gto.SphericalBasis(oracle_gto.parse_xyz(xyz), table)
    assert bs.nao == ob.nao == 28 and bs.nao_cart == 36
    for got, ref in zip(integrals.one_electron(bs), oracle_gto.one_electron_general(ob)):
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13)
    eri = integrals.two_electron_native(bs)
    rng = np.random.default_rng(1)
    nonzero = 0
    for _ in range(40):
        p, q, r, s_ = (int(x) for x in rng.integers(0, bs.nao, 4))
        ref = oracle_gto.eri_element_general(ob, p, q, r, s_)
        assert abs(ref - eri[p, q, r, s_]) < 1e-13
        nonzero += abs(ref) > 1e-6
    assert nonzero >= 20
    small = integrals.Basis(integrals.parse_geometry(xyz), {"C": [(3, (0.761,), (1.0,))], "H": [(0, (0.3,), (1.0,))]})
    np.testing.assert_allclose(integrals.two_electron_native(small), integrals.two_electron(small), rtol=0, atol=1e-14)


def _rhf_energy(m, nocc):
    s, h, eri = m["S"], m["hcore"], m["eri"]
    w, c = np.linalg.eigh(s)
    x = (c / np.sqrt(w)) @ c.T
    cmo = x @ np.linalg.eigh(x @ h @ x)[1]
    dm = 2 * cmo[:, :nocc] @ cmo[:, :nocc].T
    for _ in range(300):
        f = h + np.einsum("pqrs,rs->pq", eri, dm) - 0.5 * np.einsum("prqs,rs->pq", eri, dm)
        e = 0.5 * np.einsum("pq,pq->", dm, h + f) + m["e_nuc"]
        cmo = x @ np.linalg.eigh(x @ f @ x)[1]
        new = 2 * cmo[:, :nocc] @ cmo[:, :nocc].T
        if np.abs(new - dm).max() < 1e-9:
            break
        dm = 0.5 * (dm + new)
    return e


def _bent(sym, r, angle_deg):
    th = np.radians(angle_deg / 2)
    return f"3\n\n{sym} 0 0 0\nH 0 {r * np.sin(th)} {r * np.cos(th)}\nH 0 {-r * np.sin(th)} {r * np.cos(th)}"


def test_basis_tables_reproduce_literature_hartree_fock_energies():
    """Pins the 6-31G / 6-31G* / cc-pVDZ tables (and the d-shell code) to published closed-shell
    Hartree-Fock energies (Hehre, Radom, Schleyer, Pople, *Ab Initio Molecular Orbital Theory*, 1986,
    at the optimised geometries of the same level; cc-pVDZ water at the experimental geometry,
    Dunning 1989 / CCCBDB): every digit printed there is reproduced."""
    a = 1.082 / np.sqrt(3)
    ch4 = f"5\n\nC 0 0 0\nH {a} {a} {a}\nH {-a} {-a} {a}\nH {-a} {a} {-a}\nH {a} {-a} {-a}"
    r, al = 0.991, np.radians(116.1)
    sb = 2 / np.sqrt(3) * np.sin(al / 2)
    nh3 = "4\n\nN 0 0 0\n" + "\n".join(
        f"H {r * sb * np.cos(p)} {r * sb * np.sin(p)} {-r * np.sqrt(1 - sb * sb)}" for p in (0, 2 * np.pi / 3, 4 * np.pi / 3))
    cases = [
        (_bent("O", 0.9496, 111.55), "6-31g", False, -75.98536),
        (ch4, "6-31g", False, -40.18055),
        (nh3, "6-31g", False, -56.16552),
        (_bent("O", 0.9473, 105.5), "6-31g*", True, -76.01075),   # Pople's 6-31G* carries six Cartesian d functions
        (_bent("O", 0.9572, 104.52), "cc-pvdz", False, -76.02680),
        (_bent("O", 0.9572, 104.52), "cc-pvtz", False, -76.05717),  # [4s3p2d1f | 3s2p1d], 58 functions
    ]
    for xyz, basis, cart, literature in cases:
        m = integrals.molecule_integrals(xyz, basis, cart=cart)
        assert abs(_rhf_energy(m, 5) - literature) < 6e-6, (basis, literature)
    # the hydrogen atom in Dunning's sets (one electron: the lowest eigenvalue of hcore), N2 for nitrogen
    for basis, literature in (("cc-pvdz", -0.499278), ("cc-pvtz", -0.499810)):
        m = integrals.molecule_integrals("1\n\nH 0 0 0", basis)
        w, c = np.linalg.eigh(m["S"])
        x = (c / np.sqrt(w)) @ c.T
        assert abs(np.linalg.eigvalsh(x @ m["hcore"] @ x)[0] - literature) < 1e-6
    assert abs(_rhf_energy(integrals.molecule_integrals("2\n\nN 0 0 0\nN 0 0 1.0977", "cc-pvdz"), 7) - (-108.9541)) < 5e-5


@pytest.fixture()
def be():
    return OracleBackend()


@pytest.mark.parametrize("projector", ["mu", "huzinaga"])
def test_driver_on_builtin_provider_hf_in_hf_water(be, projector):
    """nbed(config) with nothing injected but the backend: built-in integrals, the product's own
    global mean field.  HF-in-HF embedding is exact, and the global energy is the literal value
    the reference's test asserts for water / STO-3G UHF."""
    g = load_golden("water_sto3g")
    cfg = NbedConfig(geometry=WATER_XYZ, n_active_atoms=2, basis="STO-3G", xc_functional="hf",
                     projector=projector, convergence=1e-10, max_hf_cycles=100, max_dft_cycles=100,
                     virtual_localization="cl")
    drv = nbed(cfg, backend=be)
    assert isinstance(drv.provider, BuiltinHFProvider)
    res = drv.mu if projector == "mu" else drv.huzinaga
    e_global = drv._global_ks.e_tot
    assert abs(e_global - float(g["ref_uhf_e_tot"])) < 1e-8
    assert abs(drv.e_nuc - float(g["ref_e_nuc"])) < 1e-12
    assert res["scf"].converged
    assert abs(res["e_rhf"] - e_global) < 2e-6
    assert abs(drv.e_act + drv.e_env + drv.two_e_cross + drv.e_nuc - e_global) < 1e-8


def test_builtin_provider_refuses_what_it_does_not_cover():
    from nbed_amd.exceptions import NbedDriverError

    assert BuiltinHFProvider.supports(NbedConfig(geometry=WATER_XYZ, n_active_atoms=2, basis="STO-3G",
                                                 xc_functional="b3lyp"))  # nbed_amd.xc has B3LYP
    cfg = NbedConfig(geometry=WATER_XYZ, n_active_atoms=2, basis="STO-3G", xc_functional="pbe0",
                     projector="mu", convergence=1e-8)
    assert not BuiltinHFProvider.supports(cfg)
    assert BuiltinHFProvider.supports(NbedConfig(geometry=WATER_XYZ, n_active_atoms=2, basis="cc-pVDZ",
                                                 xc_functional="b3lyp"))
    assert not BuiltinHFProvider.supports(NbedConfig(geometry=WATER_XYZ, n_active_atoms=2, basis="def2-SVP",
                                                     xc_functional="b3lyp"))
    from nbed_amd.driver import NbedDriver

    with pytest.raises(NbedDriverError):
        NbedDriver(cfg, backend=OracleBackend()).provider


def test_xc_quadrature_and_functional_derivatives():
    """nbed_amd.xc: the grid integrates the density to the electron count and the AO overlap to S;
    v_xc is the derivative of E_xc (finite differences along a random symmetric direction); LDA
    exchange of the hydrogen 1s density (spin polarised) against its closed form."""
    from nbed_amd import integrals, xc

    geom = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"
    atoms = integrals.parse_geometry(geom)
    basis = integrals.Basis(atoms, "sto-3g")
    ints = integrals.molecule_integrals(geom, "sto-3g", "angstrom")
    prov = xc.XCProvider(atoms, basis, "b3lyp", n_rad=60, n_theta=18)
    s_num = prov.ao.T @ (prov.ao * prov.weights[:, None])
    np.testing.assert_allclose(s_num, ints["S"], rtol=0, atol=5e-7)
    w, c = np.linalg.eigh(ints["S"])
    x = (c / np.sqrt(w)) @ c.T
    e, u = np.linalg.eigh(x @ ints["hcore"] @ x)
    cmo = x @ u
    dm = np.stack([cmo[:, :5] @ cmo[:, :5].T, cmo[:, :4] @ cmo[:, :4].T])  # open shell on purpose
    exc, vxc = prov(dm)
    assert abs(prov.nelec_last - 9.0) < 1e-5
    np.testing.assert_allclose(vxc, vxc.transpose(0, 2, 1), rtol=0, atol=1e-12)
    rng = np.random.default_rng(3)
    d = rng.normal(size=(2, 7, 7)) * 1e-5  # (the central difference's own error is cubic in the step)
    d = d + d.transpose(0, 2, 1)
    fd = (prov(dm + d)[0] - prov(dm - d)[0]) / 2.0
    assert abs(fd - np.einsum("xij,xji->", vxc, d)) < 2e-9
    for name in ("lda", "hf"):
        p2 = xc.XCProvider(atoms, basis, name, n_rad=40, n_theta=12)
        e2, v2 = p2(dm)
        assert (name == "hf") == (e2 == 0.0 and not np.any(v2))
    import torch

    # VWN5 (PySCF's "vwn") reproduces the Ceperley-Alder energies it was fitted to (Hartree per electron):
    # paramagnetic rs = 1, 2, 5: -0.0600, -0.0448, -0.0282; ferromagnetic rs = 1, 2: -0.0316, -0.0239
    for rs, zeta, ca in ((1, 0, -0.0600), (2, 0, -0.0448), (5, 0, -0.0282), (1, 1, -0.0316), (2, 1, -0.0239)):
        rho = 3.0 / (4.0 * np.pi * rs**3)
        ra = torch.tensor([rho * (1 + zeta) / 2 + 1e-300], dtype=torch.float64)
        rb = torch.tensor([rho * (1 - zeta) / 2 + 1e-300], dtype=torch.float64)
        assert abs(float(xc._vwn5(torch, ra, rb)) / rho - ca) < 1.2e-4
    assert xc.hybrid_fraction("b3lyp") == 0.2 and xc.hybrid_fraction("HF") == 1.0 and xc.hybrid_fraction("lda,vwn") == 0.0
    with pytest.raises(ValueError):
        xc.hybrid_fraction("pbe0")
    # closed form: E_x^LDA of rho = e^{-2r}/pi, all spin up, is -(3/2)(3/4pi)^(1/3) int rho^(4/3)
    #            = -(3/2)(3/(4 pi))^(1/3) pi^(-4/3) 4 pi 2 / (8/3)^3
    h_atom = [("H", np.zeros(3))]
    pts, wts = xc.build_grid(h_atom, n_rad=80, n_theta=8)
    import torch

    rho = torch.tensor(np.exp(-2.0 * np.linalg.norm(pts, axis=1)) / np.pi)
    z = torch.zeros_like(rho)
    ex = float((torch.tensor(wts) * xc.energy_density("slater", rho, z + 1e-30, z, z, z)).sum())
    exact = -1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3.0) * np.pi ** (-4.0 / 3.0) * 8.0 * np.pi / (8.0 / 3.0) ** 3
    assert abs(ex - exact) < 1e-6


def test_xc_default_grid_is_the_treutler_ahlrichs_lebedev_construction():
    """The default quadrature (``scheme="lebedev"``, level 3): Treutler-Ahlrichs M4 radial shells (50 for H, 75 for
    Li-Ne), Lebedev rules pruned NWChem-style (50 / 86 / 266 / 302 / 266 points from the nucleus outwards -- the
    266-point rule carries a negative weight, which must survive), Becke cells with Treutler's sqrt-radius size
    adjustment (the cell of the larger atom is the larger one).  It integrates the density and the overlap as the
    product grid does, with a tenth of the points, and gives the same exchange-correlation energy to the few 1e-6
    the reference's own grid level is good for."""
    from nbed_amd import integrals, xc

    # shell structure
    r, w = xc._ta_radial(75, xc._TA_XI["O"])
    assert r.shape == (75,) and np.all(np.diff(r) > 0) and np.all(w > 0)
    assert abs(np.sum(w * np.exp(-r * r)) - np.sqrt(np.pi) / 4.0) < 1e-12  # int r^2 exp(-r^2) dr
    degs = xc._nwchem_pruned_degrees("O", r, 29)
    assert sorted(set(int(d) for d in degs)) == [11, 15, 27, 29] and int(degs[0]) == 11 and int(degs[-1]) == 27
    assert [xc._lebedev(d)[1].size for d in (11, 15, 27, 29)] == [50, 86, 266, 302]
    assert xc._lebedev(27)[1].min() < 0.0
    o_only = xc.build_grid([("O", np.zeros(3))])
    assert o_only[0].shape[0] == sum(xc._lebedev(int(d))[1].size for d in degs)  # no point dropped, negative weights kept
    assert abs(np.sum(o_only[1] * np.exp(-np.sum(o_only[0] ** 2, axis=1))) - np.pi ** 1.5) < 1e-10
    # the molecule
    geom = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"
    atoms = integrals.parse_geometry(geom)
    basis = integrals.Basis(atoms, "sto-3g")
    ints = integrals.molecule_integrals(geom, "sto-3g", "angstrom")
    leb = xc.XCProvider(atoms, basis, "b3lyp", device="cpu")
    prod = xc.XCProvider(atoms, basis, "b3lyp", n_rad=96, n_theta=28, device="cpu")
    assert leb.points.shape[0] < 0.1 * prod.points.shape[0] and 30000 < leb.points.shape[0] < 36000
    s_num = leb.ao.T @ (leb.ao * leb.weights[:, None])
    np.testing.assert_allclose(s_num, ints["S"], rtol=0, atol=2e-6)
    w_, c = np.linalg.eigh(ints["S"])
    x = (c / np.sqrt(w_)) @ c.T
    e, u = np.linalg.eigh(x @ ints["hcore"] @ x)
    cmo = x @ u
    dm = np.stack([cmo[:, :5] @ cmo[:, :5].T, cmo[:, :5] @ cmo[:, :5].T])
    (e_l, v_l), (e_p, v_p) = leb(dm), prod(dm)
    assert abs(leb.nelec_last - 10.0) < 2e-5 and abs(e_l - e_p) < 1e-5
    np.testing.assert_allclose(v_l, v_p, rtol=0, atol=1e-3)  # (matrix elements of v_xc are what is grid sensitive)
    # Becke cells with Treutler's adjustment: the larger atom (O, Bragg radius 0.60 A against 0.35) gets the larger
    # cell -- of the points close to the O-H midpoint, those centred on O carry more weight than those centred on H
    n_o = sum(xc._lebedev(int(d))[1].size for d in xc._nwchem_pruned_degrees("O", r, 29))
    mid = 0.5 * (atoms[0][1] + atoms[1][1])
    near = np.linalg.norm(leb.points - mid, axis=1) < 0.35
    own_o = np.arange(leb.points.shape[0]) < n_o
    # (weights include r^2 dr: compare the cell SHARES, i.e. weights divided by the single-atom weights)
    o_alone = xc.build_grid([atoms[0]])
    share_o = leb.weights[:n_o] / o_alone[1]
    assert share_o[near[:n_o]].mean() > 0.5
    with pytest.raises(ValueError):
        xc.build_grid(atoms, scheme="nope")
