import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'tools'); sys.path.insert(0,'.')
from nbed_amd import NbedConfig, nbed
from nbed_amd.driver import BuiltinHFProvider
from nbed_amd.backend import HipBackend
from molecules import octane_xyz
be=HipBackend()
ch3 = "4\n\nC 0 0 0\nH 1.079 0 0\nH -0.5395 0.9344 0\nH -0.5395 -0.9344 0"
cfg = NbedConfig(geometry=ch3, n_active_atoms=2, basis="cc-pvtz", xc_functional="b3lyp", charge=1, convergence=1e-8,
                 projector="both", max_hf_cycles=200, max_dft_cycles=200, virtual_localization="cl")
drv = nbed(cfg, provider=BuiltinHFProvider(be), backend=be, hamiltonian_format="spatial")
print("CH3+", repr(drv._global_ks.e_tot), repr(drv.mu["e_rhf"]), repr(drv.huzinaga["e_rhf"]), repr(drv.huzinaga["classical_energy"]))
cfg = NbedConfig(geometry=octane_xyz(), n_active_atoms=4, basis="6-31g*", xc_functional="b3lyp", convergence=1e-8,
                 projector="both", max_hf_cycles=100, max_dft_cycles=100, localization="spade", virtual_localization="cl", max_shells=4)
for grid in (None, ("lebedev", 4), (96, 28)):
    prov = BuiltinHFProvider(be, xc_grid=grid)
    drv = nbed(cfg, provider=prov, backend=be, hamiltonian_format="spatial")
    print("octane", grid, repr(drv._global_ks.e_tot), repr(drv.mu["e_rhf"]), repr(drv.huzinaga["e_rhf"]), prov._xc_provider(cfg,"b3lyp").points.shape)
