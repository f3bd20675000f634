"""NbedDriver end to end on CPU (checker backend + synthetic provider): the product's
orchestration against the oracle's restatement of the same driver arithmetic, and against the
golden vectors of the reference's post_embed / _delete_spin_environment / _env_projector."""

import numpy as np
import pytest

from conftest import canon_sign, load_golden
from oracle import embed as o_embed
from oracle import localize as o_loc
from oracle import synth
from oracle.pyscf_like import ToyMol, ToyUHF
from oracle_backend import OracleBackend
from synthetic_provider import SyntheticProvider

from nbed_amd import NbedConfig, NbedDriver, nbed
from nbed_amd.config import ProjectorTypes
from nbed_amd.exceptions import NbedDriverError
from nbed_amd.localizers import LocalizedSystem
from nbed_amd.scf import GpuUHF, Mole

GEOM = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"


def config(**kw):
    base = dict(geometry=GEOM, n_active_atoms=1, basis="synthetic", xc_functional="none", convergence=1e-9,
                max_hf_cycles=80)
    base.update(kw)
    return NbedConfig(**base)


@pytest.fixture()
def be():
    return OracleBackend()


def reference_flow(provider, cfg, projector):
    """The same flow written with the oracle's functions (numpy)."""
    ks = provider.global_ks(cfg)
    ls, _ = o_loc.spade_localize(ks.mo_coeff, ks.mo_occ, provider.S, provider.n_act_aos)
    v_pot = np.asarray(ks.get_veff(dm=ls.dm_active + ls.dm_enviro)) - np.asarray(ks.get_veff(dm=ls.dm_active))
    nelec = (len(ls.active_mo_inds[0]), len(ls.active_mo_inds[1]))
    mf = ToyUHF(ToyMol(provider.nao, nelec, e_nuc=provider.e_nuc), provider.S, provider.h, provider.eri)
    mf.conv_tol, mf.max_cycle = cfg.convergence, cfg.max_hf_cycles
    if projector == "mu":
        scf, v_emb = o_embed.mu_embed(mf, provider.S, ls.dm_enviro, v_pot, cfg.mu_level_shift)
    else:
        scf, v_emb = o_embed.huzinaga_embed(mf, v_pot, ls.dm_enviro)
    return ls, v_pot, scf, v_emb


@pytest.mark.parametrize("projector", ["huzinaga", "mu"])
def test_embed_matches_oracle_flow(be, projector):
    prov = SyntheticProvider(14, (5, 5), 5)
    cfg = config(projector=projector, virtual_localization="disable")
    drv = nbed(cfg, provider=prov, backend=be)
    res = drv.huzinaga if projector == "huzinaga" else drv.mu
    ls, v_pot, scf, v_emb = reference_flow(prov, cfg, projector)

    np.testing.assert_array_equal(drv.localized_system.active_mo_inds, ls.active_mo_inds)
    np.testing.assert_allclose(drv.localized_system.dm_enviro, ls.dm_enviro, rtol=0, atol=1e-10)
    np.testing.assert_allclose(drv.embedding_potential, v_pot, rtol=0, atol=1e-10)
    assert bool(res["scf"].converged) and bool(scf.converged)
    # mu = 1e6 amplifies rounding: the shifted operator has eigenvalues of 1e6, both sides lose ~1e-10
    tol = 1e-8 if projector == "huzinaga" else 1e-6
    np.testing.assert_allclose(res["scf"].e_tot, scf.e_tot, rtol=0, atol=tol)
    np.testing.assert_allclose(res["v_emb"], v_emb, rtol=1e-12, atol=tol)

    n_env = ls.c_enviro.shape[-1]
    assert res["scf"].mo_coeff.shape == (2, prov.nao, prov.nao - n_env)
    o_res = o_embed.post_embed(scf, v_emb, projector, ls.dm_active, ls.enviro_mo_inds,
                               o_embed.env_projector(prov.S, ls.dm_enviro), drv.e_env, drv.two_e_cross, drv.e_nuc,
                               eri=prov.eri)
    for key in ("correction", "beta_correction", "e_rhf", "classical_energy", "hf_emb"):
        np.testing.assert_allclose(res[key], o_res[key], rtol=0, atol=max(tol, 1e-8), err_msg=key)
    np.testing.assert_allclose(res["mo_energies_emb_post_del"], o_res["mo_energies_emb_post_del"], rtol=0, atol=1e-5)
    const, h1, h2 = res["second_quantised"]
    nq = 2 * (prov.nao - n_env)
    assert h1.shape == (nq, nq) and h2.shape == (nq,) * 4
    assert const == res["classical_energy"] == drv.classical_energy
    assert drv.embedded_scf is res["scf"]
    if projector == "huzinaga":  # gauge-free comparison of the Hamiltonian: |coefficients|
        np.testing.assert_allclose(np.abs(h1), np.abs(o_res["second_quantised"][1]), rtol=0, atol=1e-6)


def test_projectors_agree(be):
    """mu-shift and Huzinaga embedding give the same embedded energy (tests/test_driver.py:142-152)."""
    prov = SyntheticProvider(14, (5, 5), 5)
    drv = nbed(config(projector="both", virtual_localization="cl"), provider=prov, backend=be)
    assert drv.mu["scf"].converged and drv.huzinaga["scf"].converged
    assert abs(drv.mu["e_rhf"] - drv.huzinaga["e_rhf"]) < 1e-5
    assert isinstance(drv.embedded_scf, tuple) and isinstance(drv.classical_energy, tuple)
    for res in (drv.mu, drv.huzinaga):
        assert set(res) >= {"scf", "v_emb", "mo_energies_emb_pre_del", "mo_energies_emb_post_del", "correction",
                            "beta_correction", "cl", "e_rhf", "classical_energy", "hf_emb", "second_quantised"}
        assert res["cl"].shells[0][-1] == res["scf"].mo_coeff.shape[-1]


def test_post_embed_matches_reference_golden(be):
    """Same inputs as tests/golden/post_embed_huzinaga_n12.npz (written by the reference's
    _huzinaga_embed + post_embed)."""
    g = load_golden("post_embed_huzinaga_n12")
    n = int(g["nao"])
    drv = NbedDriver(config(projector="huzinaga", virtual_localization="disable", convergence=float(g["conv_tol"]),
                            max_hf_cycles=int(g["max_cycle"])), provider=object(), backend=be)

    class KS:
        def get_ovlp(self):
            return g["S"]

    drv.__dict__["_global_ks"] = KS()
    drv.localized_system = LocalizedSystem(np.array([np.arange(3), np.arange(3)]),
                                           np.array([np.arange(3, 5), np.arange(3, 5)]),
                                           g["c_active"], g["c_enviro"], g["c_loc_occ"], backend=be)
    drv.e_env, drv.two_e_cross, drv.e_nuc = float(g["e_env"]), float(g["two_e_cross"]), float(g["e_nuc"])
    mf = GpuUHF(Mole(n, tuple(g["nelec"]), e_nuc=float(g["e_nuc"])), g["S"], g["hcore"], synth.eri_dense(n), backend=be)
    mf.max_cycle, mf.conv_tol = int(g["max_cycle"]), float(g["conv_tol"])
    scf, v_emb = drv._huzinaga_embed(mf, g["V_emb"], drv.localized_system, None)
    res = drv.post_embed(scf, v_emb, ProjectorTypes.HUZ)
    np.testing.assert_allclose(v_emb, g["v_emb"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(scf.e_tot, g["e_tot"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res["mo_energies_emb_pre_del"], g["mo_energies_pre"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res["mo_energies_emb_post_del"], g["mo_energies_post"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(res["scf"].mo_occ, g["mo_occ_post"])
    np.testing.assert_allclose(canon_sign(res["scf"].mo_coeff), g["mo_coeff_post_canon"], rtol=0, atol=1e-6)
    for k in ("correction", "beta_correction", "e_rhf", "classical_energy", "hf_emb"):
        np.testing.assert_allclose(res[k], g[k], rtol=0, atol=1e-9)
    const, h1, h2 = res["second_quantised"]
    np.testing.assert_allclose(const, g["const"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.abs(h1), g["h1_abs"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(np.abs(h2), g["h2_abs"], rtol=0, atol=1e-7)


def test_delete_environment_and_projector_match_reference(be):
    g = load_golden("delete_environment")
    drv = NbedDriver(config(), provider=object(), backend=be)
    for key, ptype in [("huz", ProjectorTypes.HUZ), ("mu", ProjectorTypes.MU)]:
        for x in range(2):
            c, e, o = drv._delete_spin_environment(ptype, int(g["n_env"]), g["mo_coeff"][x], g["mo_energy"][x],
                                                   g["mo_occ"][x], g["projector"][x])
            np.testing.assert_array_equal(c, g[f"{key}_coeff_{x}"])
            np.testing.assert_array_equal(e, g[f"{key}_energy_{x}"])
            np.testing.assert_array_equal(o, g[f"{key}_occ_{x}"])
    g = load_golden("env_projector")

    class KS:
        def get_ovlp(self):
            return g["S"]

    drv.__dict__["_global_ks"] = KS()
    drv.localized_system = type("LS", (), {"dm_enviro": g["dm_enviro"]})()
    np.testing.assert_allclose(drv._env_projector, g["projector"], rtol=0, atol=1e-12)


def test_driver_errors(be):
    with pytest.raises(NotImplementedError, match="PAO"):
        NbedDriver(config(virtual_localization="pao"), provider=SyntheticProvider(14, (5, 5), 5), backend=be).embed()
    with pytest.raises(NotImplementedError):
        nbed(config(localization="pm"), provider=SyntheticProvider(14, (5, 5), 5), backend=be)
    try:
        import pyscf  # noqa: F401
    except ImportError:
        # CCSD without PySCF: small active spaces go through nbed_amd.ccsd (the correlation energy lowers the
        # embedded energy), larger ones are refused loudly
        drv = nbed(config(run_ccsd_emb=True, virtual_localization="disable"), provider=SyntheticProvider(14, (5, 5), 5),
                   backend=be)
        res = drv.mu if drv.mu is not None else drv.huzinaga
        assert res["e_ccsd"] < res["e_rhf"] and res["e_rhf"] - res["e_ccsd"] < 1.0
        with pytest.raises(NbedDriverError, match="PySCF"):
            nbed(config(run_ccsd_emb=True, virtual_localization="disable"), provider=SyntheticProvider(24, (5, 5), 5),
                 backend=be)
    with pytest.raises(NotImplementedError):  # a provider without local_ks cannot do DFT-in-DFT
        nbed(config(run_dft_in_dft=True, virtual_localization="disable"), provider=SyntheticProvider(14, (5, 5), 5),
             backend=be)
    try:
        import pyscf  # noqa: F401
    except ImportError:
        with pytest.raises(NbedDriverError, match="PySCF"):
            NbedDriver(config(), backend=be).embed()


def test_qmmm_field_reaches_the_embedded_scf_or_is_refused(be):
    """QM/MM (nbed/driver.py:171-180, 246-253): the point-charge field must be in the embedded
    object's hcore and energy_nuc, consistently with the global one; a provider whose local_hf()
    cannot add it is refused instead of being run without the field."""
    mm = dict(mm_coords=[[3.0, 0.0, 0.0]], mm_charges=[0.4], mm_radii=[1.0])
    cfg = config(projector="huzinaga", virtual_localization="disable", **mm)

    plain = SyntheticProvider(14, (5, 5), 5)
    drv = NbedDriver(cfg, provider=plain, backend=be)
    assert drv.run_qmmm
    with pytest.raises((NbedDriverError, TypeError)):
        drv.embed()

    class FieldProvider(SyntheticProvider):
        """Adds a fixed one-electron 'point-charge' potential and a constant MM-nuclear energy."""

        v_mm = 0.01 * synth.sym_matrix(synth.STREAM_MISC + 20, 14)
        e_mm = 0.37
        seen = []

        def global_ks(self, config, run_qmmm=False):
            self.seen.append(("global", run_qmmm))
            h0, e0 = self.h, self.e_nuc
            if run_qmmm:
                self.h, self.e_nuc = h0 + self.v_mm, e0 + self.e_mm
            try:
                return super().global_ks(config)
            finally:
                self.h, self.e_nuc = h0, e0

        def local_hf(self, config, embedded_mol, backend=None, run_qmmm=False):
            self.seen.append(("local", run_qmmm))
            mf = GpuUHF(embedded_mol, self.S, self.h + (self.v_mm if run_qmmm else 0.0), self.eri, backend=backend)
            if run_qmmm:
                e_nuc = self.e_nuc + self.e_mm
                mf.energy_nuc = lambda *a: e_nuc
            return mf

    prov = FieldProvider(14, (5, 5), 5)
    drv = nbed(cfg, provider=prov, backend=be)
    assert ("global", True) in prov.seen and ("local", True) in prov.seen
    scf = drv.huzinaga["scf"]
    np.testing.assert_allclose(np.asarray(scf._h_h), prov.h + prov.v_mm, rtol=0, atol=0)  # field in the local hcore
    assert abs(scf.energy_nuc() - drv.e_nuc) < 1e-14  # ... and the same MM-nuclear term on both sides
    # HF-in-HF with the field everywhere stays exact: embedded energy == global energy
    assert abs(drv.huzinaga["e_rhf"] - drv._global_ks.e_tot) < 1e-7
    drv0 = nbed(config(projector="huzinaga", virtual_localization="disable"), provider=FieldProvider(14, (5, 5), 5),
                backend=be)
    assert abs(drv0.huzinaga["e_rhf"] - drv.huzinaga["e_rhf"]) > 1e-4  # the field is not a no-op


def test_dft_in_dft_matches_reference_golden(be):
    """nbed/driver.py:1138-1231 dft_in_dft (Huzinaga projector): the product's function with a
    GpuUKS local Kohn-Sham object (toy hybrid functional, exact-exchange fraction from the J/K
    kernels) against the reference's own run."""
    from nbed_amd.driver import dft_in_dft
    from nbed_amd.scf import GpuUKS

    g = load_golden("dft_in_dft_huzinaga_n12")
    n = int(g["nao"])

    class Provider:
        def local_ks(self, cfg, mol, xc, backend=None):
            ks = GpuUKS(Mole(n, tuple(g["nelec"]), e_nuc=float(g["e_nuc"])), g["S"], g["hcore"], synth.eri_dense(n),
                        backend=backend, xc=xc, hyb=float(g["hyb"]))
            ks.max_cycle = int(g["max_cycle"])
            return ks

        def build_mol(self, cfg):
            return Mole(n, tuple(g["nelec"]), e_nuc=float(g["e_nuc"]))

    drv = NbedDriver(config(projector="huzinaga", virtual_localization="disable", convergence=float(g["conv_tol"]),
                            max_hf_cycles=int(g["max_cycle"])), provider=Provider(), backend=be)

    class KS:
        xc = "toy-hybrid"

        def get_ovlp(self):
            return g["S"]

        def energy_nuc(self):
            return float(g["e_nuc"])

    drv.__dict__["_global_ks"] = KS()
    drv.localized_system = LocalizedSystem(np.array([np.arange(3), np.arange(3)]),
                                           np.array([np.arange(3, 5), np.arange(3, 5)]),
                                           g["c_active"], g["c_enviro"], g["c_loc_occ"], backend=be)
    drv.e_env, drv.two_e_cross, drv.e_nuc = float(g["e_env"]), float(g["two_e_cross"]), float(g["e_nuc"])
    drv.embedding_potential = g["V_emb"]
    res = dft_in_dft(drv, ProjectorTypes.HUZ)
    assert bool(res["scf_dft"].converged) == bool(g["converged"])
    np.testing.assert_allclose(res["v_emb_dft"], g["v_emb_dft"], rtol=0, atol=1e-8)
    for k in ("dft_correction", "dft_correction_beta", "e_dft_in_dft", "emb_dft"):
        np.testing.assert_allclose(res[k], g[k], rtol=0, atol=1e-8, err_msg=k)
    np.testing.assert_allclose(res["scf_dft"].mo_energy, g["mo_energy_post"], rtol=0, atol=1e-8)
    np.testing.assert_array_equal(res["scf_dft"].mo_occ, g["mo_occ_post"])
    np.testing.assert_allclose(canon_sign(res["scf_dft"].mo_coeff), g["mo_coeff_post_canon"], rtol=0, atol=1e-6)


def test_consumers_are_importable_and_fail_loudly_without_pyscf(be):
    """run_emb_fci / run_emb_ccsd (nbed/driver.py:1044-1135) delegate to PySCF's solvers: importable from
    the package root like the reference's, NbedDriverError (not a silent skip) without PySCF."""
    import nbed_amd
    from nbed_amd.driver import run_emb_ccsd, run_emb_fci

    assert nbed_amd.run_emb_fci is run_emb_fci and nbed_amd.run_emb_ccsd is run_emb_ccsd
    try:
        import pyscf  # noqa: F401
    except ImportError:
        for fn in (run_emb_fci, run_emb_ccsd):
            with pytest.raises(NbedDriverError):
                fn(object())


def test_nbed_config_input_forms(be, tmp_path):
    """tests/test_embed.py:10-41 -- the forms ``nbed()`` accepts: a config object, keyword overrides on
    top of it, a path to a .json file, bare keyword arguments, ``config=None``, an object that is not a config
    (ignored in favour of the keywords), and a pydantic ValidationError when a required field is missing."""
    import json

    from pydantic import ValidationError

    from nbed_amd.config import parse_config

    base = config(virtual_localization="disable")
    prov = SyntheticProvider(14, (5, 5), 5)
    assert isinstance(nbed(base, provider=prov, backend=be), NbedDriver)
    assert nbed(base, provider=prov, backend=be, n_active_atoms=1).config.n_active_atoms == 1
    args = base.model_dump()
    path = tmp_path / "config.json"
    path.write_text(json.dumps(base.model_dump(mode="json")))
    assert parse_config(str(path)) == base and parse_config(path) == base
    assert isinstance(nbed(str(path), provider=prov, backend=be), NbedDriver)
    assert isinstance(nbed(provider=prov, backend=be, **args), NbedDriver)
    assert parse_config(None, **args) == base
    assert parse_config(["a", "list"], **args) == base
    args.pop("geometry")
    with pytest.raises(ValidationError):
        parse_config(None, **args)


def test_geometry_helpers(tmp_path):
    """tests/test_utils.py:32-58: active atoms first, tab separated, the file under molecular_structures/."""
    from nbed_amd.utils import build_ordered_xyz_string, save_ordered_xyz_file

    water = {0: ("O", (0, 0, 0)), 1: ("H", (0.2774, 0.8929, 0.2544)), 2: ("H", (0.6068, -0.2383, -0.7169))}
    o_first = "3\n \nO\t0\t0\t0\nH\t0.2774\t0.8929\t0.2544\nH\t0.6068\t-0.2383\t-0.7169\n"
    assert build_ordered_xyz_string(water, [0]) == o_first
    assert build_ordered_xyz_string(water, [1, 2]) == "3\n \nH\t0.2774\t0.8929\t0.2544\nH\t0.6068\t-0.2383\t-0.7169\nO\t0\t0\t0\n"
    path = save_ordered_xyz_file("water_test", water, [0], save_location=tmp_path)
    assert path.read_text() == o_first and path.parent.name == "molecular_structures"
    with pytest.raises(ValueError):
        build_ordered_xyz_string(water, [5])
