"""The CPU oracle against the golden vectors produced by the reference itself.

Every fixture under tests/golden was written by tests/golden/make_golden.py,
which ran the reference's own functions (file:line in MANIFEST.json).  These
tests pin oracle/ to them; the GPU parity tests then compare the HIP path with
the oracle and with the same fixtures.
"""

import json

import numpy as np
import pytest

from conftest import GOLDEN, canon_sign, load_golden
from oracle import embed, hamiltonian, huzinaga, localize, synth
from oracle.pyscf_like import ToyMol, ToyRHF, ToyUHF, ToyUKS

TOL = dict(rtol=0, atol=1e-11)


def test_manifest_lists_every_fixture():
    manifest = json.loads((GOLDEN / "MANIFEST.json").read_text())
    files = sorted(p.stem for p in GOLDEN.glob("*.npz"))
    assert files == sorted(manifest)
    for name, meta in manifest.items():
        assert meta["reference"], name


def test_huzinaga_operator():
    g = load_golden("huzinaga_operator")
    out = huzinaga.get_huzinaga_operator(g["fock"], g["dm_occ_S"], g["dm_virt_S"])
    np.testing.assert_allclose(out, g["out_3d"], **TOL)
    out = huzinaga.get_huzinaga_operator(g["fock"], g["dm_occ_S"], np.zeros_like(g["dm_occ_S"]))
    np.testing.assert_allclose(out, g["out_3d_novirt"], **TOL)
    out = huzinaga.get_huzinaga_operator(g["fock"][0], g["dm_occ_S"][0], g["dm_virt_S"][0])
    np.testing.assert_allclose(out, g["out_2d"], **TOL)


@pytest.mark.parametrize("tag", ["uhf_n12_nodiis", "uhf_n12_diis", "uhf_n24_diis_open", "uhf_n24_nodiis_open"])
def test_huzinaga_scf_uhf(tag):
    g = load_golden(f"huzinaga_scf_{tag}")
    n = int(g["nao"])
    eri = synth.eri_dense(n)
    mf = ToyUHF(ToyMol(n, tuple(g["nelec"])), g["S"], g["hcore"], eri)
    mf.max_cycle, mf.conv_tol = int(g["max_cycle"]), float(g["conv_tol"])
    c, e, d, hz, conv = huzinaga.huzinaga_scf(mf, g["V_emb"], g["D_env"], use_DIIS=bool(g["use_DIIS"]))
    assert conv == bool(g["conv"])
    # the reference takes S^-1/2 from scipy's Schur-based fractional_matrix_power, the
    # oracle from eigh: same matrix to rounding, amplified a little by the SCF iterations
    np.testing.assert_allclose(e, g["mo_energy"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(d, g["dm"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(hz, g["huz_op"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(canon_sign(c), g["mo_coeff_canon"], rtol=0, atol=1e-7)


def test_huzinaga_scf_exact_power_matches_tighter():
    g = load_golden("huzinaga_scf_uhf_n12_nodiis")
    n = int(g["nao"])
    mf = ToyUHF(ToyMol(n, tuple(g["nelec"])), g["S"], g["hcore"], synth.eri_dense(n))
    mf.max_cycle, mf.conv_tol = int(g["max_cycle"]), float(g["conv_tol"])
    c, e, d, hz, conv = huzinaga.huzinaga_scf(mf, g["V_emb"], g["D_env"], use_DIIS=False, exact_power=True)
    np.testing.assert_allclose(e, g["mo_energy"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(d, g["dm"], rtol=0, atol=1e-12)


def test_huzinaga_scf_rhf():
    g = load_golden("huzinaga_scf_rhf_n12_diis")
    n = int(g["nao"])
    mf = ToyRHF(ToyMol(n, tuple(g["nelec"])), g["S"], g["hcore"], synth.eri_dense(n))
    mf.max_cycle, mf.conv_tol = int(g["max_cycle"]), float(g["conv_tol"])
    c, e, d, hz, conv = huzinaga.huzinaga_scf(mf, g["V_emb"], g["D_env"], use_DIIS=True)
    assert conv == bool(g["conv"])
    np.testing.assert_allclose(e, g["mo_energy"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(d, g["dm"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(hz, g["huz_op"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag", ["uks_n12_diis", "uks_n24_nodiis_open"])
def test_huzinaga_scf_uks_branch(tag):
    """The Kohn-Sham branch (huzinaga_scf.py:176-180, calculate_ks_energy :36-62) against the
    reference's own run on a toy hybrid functional (tagged veff with .ecoul/.exc)."""
    g = load_golden(f"huzinaga_scf_{tag}")
    n = int(g["nao"])
    ks = ToyUKS(ToyMol(n, tuple(g["nelec"])), g["S"], g["hcore"], synth.eri_dense(n))
    ks.hyb = float(g["hyb"])
    ks.max_cycle, ks.conv_tol = int(g["max_cycle"]), float(g["conv_tol"])
    calls = []
    orig = ks.get_veff
    ks.get_veff = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    hist = []
    c, e, d, hz, conv = huzinaga.huzinaga_scf(ks, g["V_emb"], g["D_env"], use_DIIS=bool(g["use_DIIS"]), history=hist)
    assert conv == bool(g["conv"])
    assert len(calls) == 2 * len(hist)  # the KS branch costs a second get_veff per cycle (:55)
    tol = 1e-8 if bool(g["use_DIIS"]) else 1e-9
    np.testing.assert_allclose(e, g["mo_energy"], rtol=0, atol=tol)
    np.testing.assert_allclose(d, g["dm"], rtol=0, atol=tol)
    np.testing.assert_allclose(hz, g["huz_op"], rtol=0, atol=tol)
    e_ks = huzinaga.calculate_ks_energy(ks, g["V_emb"], g["dm"], g["huz_op"])
    assert e_ks.shape == (2,)
    np.testing.assert_allclose(e_ks, g["e_ks"], rtol=0, atol=1e-11)


def test_energy_elec():
    g = load_golden("energy_elec")
    n = int(g["nao"])
    mf = ToyUHF(ToyMol(n, (5, 4)), g["S"], g["hcore3"][0], synth.eri_dense(n))
    e_elec, e_coul = huzinaga.energy_elec(mf, g["dm"], g["hcore3"], None)
    np.testing.assert_allclose(e_elec, g["e_elec"], **TOL)
    np.testing.assert_allclose(e_coul, g["e_coul"], **TOL)
    np.testing.assert_allclose(mf.scf_summary["e1"], g["e1"], **TOL)


@pytest.mark.parametrize("tag", ["n16_closed", "n16_overwrite", "n20_wide"])
def test_spade(tag):
    g = load_golden(f"spade_{tag}")
    ow = tuple(None if o < 0 else int(o) for o in g["overwrite"])
    ls, cond = localize.spade_localize(g["mo_coeff"], g["mo_occ"], g["S"], int(g["n_act_aos"]), ow)
    np.testing.assert_array_equal(ls.active_mo_inds, g["active_mo_inds"])
    np.testing.assert_array_equal(ls.enviro_mo_inds, g["enviro_mo_inds"])
    np.testing.assert_allclose(cond[0], g["sigma_a"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(cond[1], g["sigma_b"], rtol=0, atol=1e-12)
    # singular vectors carry a sign/rotation gauge: compare the projectors
    np.testing.assert_allclose(ls.dm_active, g["dm_active"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(ls.dm_enviro, g["dm_enviro"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(ls.dm_loc_occ, g["dm_loc_occ"], rtol=0, atol=1e-10)
    assert ls.c_active.shape == g["c_active"].shape
    assert ls.c_enviro.shape == g["c_enviro"].shape


def test_spade_restricted():
    g = load_golden("spade_n14_restricted")
    ls, cond = localize.spade_localize(g["mo_coeff"], g["mo_occ"], g["S"], int(g["n_act_aos"]))
    np.testing.assert_array_equal(ls.active_mo_inds, g["active_mo_inds"])
    np.testing.assert_allclose(ls.dm_active, g["dm_active"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(ls.dm_enviro, g["dm_enviro"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(cond[0], g["sigma_a"], rtol=0, atol=1e-12)


def test_spade_open_shell_raises_like_reference():
    g = load_golden("spade_n16_open_raises")
    assert str(g["raised"]).startswith("ValueError")
    n, nocc = int(g["nao"]), tuple(int(x) for x in g["nocc"])
    pr = synth.problem(n, nocc, 0)
    _, ca = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    occ = np.zeros((2, n))
    occ[0, : nocc[0]] = 1
    occ[1, : nocc[1]] = 1
    with pytest.raises(ValueError):
        localize.spade_localize(np.stack([ca, ca]), occ, pr["S"], int(g["n_act_aos"]))


def test_env_projector_and_mu_potential():
    g = load_golden("env_projector")
    p = embed.env_projector(g["S"], g["dm_enviro"])
    np.testing.assert_allclose(p, g["projector"], **TOL)
    v = embed.mu_v_emb(float(g["mu"]), p, g["V_emb"])
    np.testing.assert_allclose(v, g["v_emb"], rtol=1e-15, atol=0)


def test_delete_environment():
    g = load_golden("delete_environment")
    for key, ptype in [("huz", "huzinaga"), ("mu", "mu")]:
        for x in range(2):
            c, e, o = embed.delete_spin_environment(
                ptype, int(g["n_env"]), g["mo_coeff"][x], g["mo_energy"][x], g["mo_occ"][x], g["projector"][x]
            )
            np.testing.assert_array_equal(c, g[f"{key}_coeff_{x}"])
            np.testing.assert_array_equal(e, g[f"{key}_energy_{x}"])
            np.testing.assert_array_equal(o, g[f"{key}_occ_{x}"])


@pytest.mark.parametrize("max_shells", [4, 1])
def test_concentric(max_shells):
    g = load_golden(f"concentric_n18_shells{max_shells}")
    na = int(g["n_act_proj_aos"])
    c, shells, svals = localize.concentric_localize_spin(
        g["occ"], g["mo_coeff"], g["fock"], g["S"][:na, :na], g["S"][:na, :], na, int(g["max_shells"])
    )
    np.testing.assert_array_equal(shells, g["shells"])
    assert len(svals) == int(g["n_sigma"])
    for i, s in enumerate(svals):
        np.testing.assert_allclose(s, g[f"sigma_{i}"], rtol=0, atol=1e-11)
    assert c.shape == g["out_coeff"].shape
    # shell subspaces are gauge invariant: compare the projector of each shell
    edges = [int(np.count_nonzero(g["occ"]))] + list(shells)
    for a, b in zip(edges[:-1], edges[1:]):
        np.testing.assert_allclose(c[:, a:b] @ c[:, a:b].T, g["out_coeff"][:, a:b] @ g["out_coeff"][:, a:b].T,
                                   rtol=0, atol=1e-9)


def test_spinorb_from_spatial():
    g = load_golden("spinorb_from_spatial")
    h1, h2 = hamiltonian.spinorb_from_spatial(g["one_body"], g["two_body"])
    np.testing.assert_array_equal(h1, g["h1"])
    np.testing.assert_array_equal(h2, g["h2"])


def test_ham_build():
    g = load_golden("ham_build_n10")
    eri = synth.eri_dense(int(g["nao"]))
    const, h1, h2 = hamiltonian.build(g["mo_coeff"], g["hcore3"], eri, float(g["const"]))
    assert const == float(g["const"])
    np.testing.assert_allclose(hamiltonian.one_body_integrals(g["mo_coeff"], g["hcore3"]), g["one_body"], **TOL)
    np.testing.assert_allclose(hamiltonian.two_body_integrals(g["mo_coeff"], eri), g["two_body"], **TOL)
    np.testing.assert_allclose(h1, g["h1"], **TOL)
    np.testing.assert_allclose(h2, g["h2"], **TOL)


def test_post_embed_huzinaga():
    g = load_golden("post_embed_huzinaga_n12")
    n = int(g["nao"])
    eri = synth.eri_dense(n)
    mf = ToyUHF(ToyMol(n, tuple(g["nelec"]), e_nuc=float(g["e_nuc"])), g["S"], g["hcore"], eri)
    mf.max_cycle, mf.conv_tol = int(g["max_cycle"]), float(g["conv_tol"])
    ls = localize.LocalizedSystem(
        np.array([np.arange(3), np.arange(3)]), np.array([np.arange(3, 5), np.arange(3, 5)]),
        g["c_active"], g["c_enviro"], g["c_loc_occ"])
    scf, v_emb = embed.huzinaga_embed(mf, g["V_emb"], ls.dm_enviro)
    assert bool(scf.converged) == bool(g["converged"])
    np.testing.assert_allclose(v_emb, g["v_emb"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(scf.e_tot, g["e_tot"], rtol=0, atol=1e-9)
    res = embed.post_embed(scf, v_emb, "huzinaga", ls.dm_active, ls.enviro_mo_inds,
                           embed.env_projector(g["S"], ls.dm_enviro),
                           float(g["e_env"]), float(g["two_e_cross"]), float(g["e_nuc"]), eri=eri)
    np.testing.assert_allclose(res["mo_energies_emb_pre_del"], g["mo_energies_pre"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res["mo_energies_emb_post_del"], g["mo_energies_post"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(res["scf"].mo_occ, g["mo_occ_post"])
    for k in ("correction", "beta_correction", "e_rhf", "classical_energy", "hf_emb"):
        np.testing.assert_allclose(res[k], g[k], rtol=0, atol=1e-9)
    const, h1, h2 = res["second_quantised"]
    np.testing.assert_allclose(const, g["const"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.abs(h1), g["h1_abs"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(np.abs(h2), g["h2_abs"], rtol=0, atol=1e-7)
