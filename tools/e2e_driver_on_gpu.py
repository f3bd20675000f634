"""End-to-end NbedDriver run at bench scale on the GPU with a GPU-resident synthetic provider."""
import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from nbed_amd import NbedConfig, nbed, synth
from nbed_amd.backend import HipBackend
from nbed_amd.scf import GpuUHF, Mole
class TaggedArray(np.ndarray):
    pass

N = int(os.environ.get("E2E_N", "148")); nocc = int(os.environ.get("E2E_NOCC", "33")); nact = int(os.environ.get("E2E_NACT", "60"))
be = HipBackend()
GEOM = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"

class GpuKS(GpuUHF):
    xc = "exact-exchange"
    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
        dm = self.make_rdm1() if dm is None else np.asarray(dm)
        dm3 = np.array((dm * 0.5, dm * 0.5)) if dm.ndim == 2 else dm
        jk = self.be.to_host(self.jk_device(self.be.asarray(dm3)))
        v = (jk[0] - jk[1:]).view(TaggedArray)
        v.ecoul = 0.5 * float(np.einsum("ij,ji->", jk[0], dm3[0] + dm3[1]))
        v.exc = -0.5 * float(np.einsum("xij,xji->", jk[1:], dm3))
        return v

class Provider:
    def __init__(self):
        t = time.perf_counter()
        self.S, self.h = synth.overlap(N), synth.hcore(N)
        self.eri = be.synth_eri(N); torch.cuda.synchronize()
        print(f"integrals on device: {time.perf_counter()-t:.2f} s", flush=True)
        self.slices = [[0, 1, 0, nact], [1, 2, nact, N]]
    def build_mol(self, config):
        return Mole(N, (nocc, nocc), ao_slices=self.slices, e_nuc=1.25, atom=config.geometry, basis=config.basis)
    def global_ks(self, config):
        t = time.perf_counter()
        ks = GpuKS(Mole(N, (nocc, nocc), ao_slices=self.slices, e_nuc=1.25), self.S, self.h, self.eri, backend=be)
        ks.conv_tol, ks.max_cycle = 1e-10, 100
        ks.kernel(); torch.cuda.synchronize()
        print(f"global mean field: {ks.cycles} cycles, converged={ks.converged}, {time.perf_counter()-t:.2f} s", flush=True)
        return ks
    def local_hf(self, config, embedded_mol, backend=None):
        return GpuUHF(embedded_mol, self.S, self.h, self.eri, backend=backend)

import logging
cfg = NbedConfig(geometry=GEOM, n_active_atoms=1, basis="synthetic", xc_functional="none", convergence=1e-8,
                 max_hf_cycles=100, projector="both", virtual_localization=os.environ.get("E2E_VLOC", "disable"))
prov = Provider()
t0 = time.perf_counter()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
FMT = os.environ.get("E2E_FORMAT", "dense")  # "spatial": three unique spin blocks instead of the (2n)^4 tensor
drv = nbed(cfg, provider=prov, backend=be, hamiltonian_format=FMT)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(int(os.environ.get("E2E_NSTATS", "28")))
torch.cuda.synchronize()
print(f"NbedDriver.embed() total: {time.perf_counter()-t0:.2f} s", flush=True)
for name in ("mu", "huzinaga"):
    res = getattr(drv, name)
    sq = res["second_quantised"]
    shape = sq[2].shape if FMT == "dense" else ("spatial", sq.two_body.shape, f"{sq.nbytes / 1e9:.2f} GB")
    print(name, "e_rhf", res["e_rhf"], "classical", res["classical_energy"], "h2", shape, "converged", bool(res["scf"].converged))
print("global e_tot", drv._global_ks.e_tot)
