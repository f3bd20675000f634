"""Real-molecule known-answer tests: water / STO-3G, the reference's own test molecule.

* the global UHF energy the reference's test asserts (tests/test_driver.py:52-61, a PySCF
  number) is reproduced from scratch -- oracle integrals (oracle/gto.py) + SCF;
* HF-in-HF projection-based embedding of the same molecule is exact (e_rhf == global energy)
  for both projectors: the DFT-free analogue of tests/test_driver.py:83-88.
Run on CPU with the checker backend here; tests/test_gpu_host_suites.py re-runs them on libnbx."""

import numpy as np
import pytest

from conftest import load_golden
from oracle.pyscf_like import ToyMol, ToyUHF
from oracle_backend import OracleBackend
from synthetic_provider import SyntheticProvider

from nbed_amd import NbedConfig, nbed
from nbed_amd.scf import GpuUHF, Mole

WATER_XYZ = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"


@pytest.fixture()
def be():
    return OracleBackend()


def test_oracle_reproduces_reference_uhf_literals():
    g = load_golden("water_sto3g")
    assert abs(float(g["e_nuc"]) - float(g["ref_e_nuc"])) < 1e-12
    mf = ToyUHF(ToyMol(7, (5, 5), e_nuc=float(g["e_nuc"])), g["S"], g["T"] + g["V"], g["eri"])
    mf.conv_tol, mf.max_cycle = 1e-12, 100
    e_tot = mf.kernel()
    assert mf.converged
    # PySCF's own value is converged to its default conv_tol = 1e-9
    assert abs(e_tot - float(g["ref_uhf_e_tot"])) < 1e-8
    e_elec, e_coul = mf.energy_elec()
    assert abs(e_elec - g["ref_uhf_energy_elec"][0]) < 1e-8
    assert abs(e_coul - g["ref_uhf_energy_elec"][1]) < 1e-5  # first order in PySCF's residual density error


def test_product_scf_reproduces_reference_uhf_literals(be):
    g = load_golden("water_sto3g")
    mf = GpuUHF(Mole(7, (5, 5), e_nuc=float(g["e_nuc"])), g["S"], g["T"] + g["V"], g["eri"], backend=be)
    mf.conv_tol, mf.max_cycle = 1e-12, 100
    e_tot = mf.kernel()
    assert mf.converged
    assert abs(e_tot - float(g["ref_uhf_e_tot"])) < 1e-8
    assert abs(mf.energy_nuc() - float(g["ref_e_nuc"])) < 1e-12
    assert abs(mf.energy_elec()[0] - g["ref_uhf_energy_elec"][0]) < 1e-8
    assert mf.mo_coeff.shape == (2, 7, 7) and np.all(mf.mo_occ == [[1, 1, 1, 1, 1, 0, 0]] * 2)


@pytest.mark.parametrize("projector", ["mu", "huzinaga"])
@pytest.mark.parametrize("n_active_atoms", [1, 2])
def test_hf_in_hf_embedding_of_water_is_exact(be, projector, n_active_atoms):
    g = load_golden("water_sto3g")
    prov = SyntheticProvider.water_sto3g(g)
    if n_active_atoms == 2:
        prov.n_act_aos = prov.slices[1][3]
    cfg = NbedConfig(geometry=WATER_XYZ, n_active_atoms=n_active_atoms, basis="STO-3G", xc_functional="hf",
                     projector=projector, convergence=1e-10, max_hf_cycles=100, virtual_localization="cl")
    drv = nbed(cfg, provider=prov, backend=be)
    res = drv.mu if projector == "mu" else drv.huzinaga
    e_global = drv._global_ks.e_tot
    assert abs(e_global - float(g["ref_uhf_e_tot"])) < 1e-8
    assert res["scf"].converged
    assert abs(res["e_rhf"] - e_global) < 2e-6, (res["e_rhf"], e_global)
    # subsystem energies add up to the global energy (tests/test_driver.py:200-224)
    assert abs(drv.e_act + drv.e_env + drv.two_e_cross + drv.e_nuc - e_global) < 1e-8
    n_env = len(drv.localized_system.enviro_mo_inds[0])
    assert res["scf"].mo_coeff.shape == (2, 7, 7 - n_env)
    const, h1, h2 = res["second_quantised"]
    assert h1.shape == (2 * (7 - n_env),) * 2 and const == res["classical_energy"]
