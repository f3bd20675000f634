"""Provider for NbedDriver tests: synthetic integrals and a global mean-field standing in for
the B3LYP Kohn-Sham calculation (exchange-correlation set to exact exchange, E_xc folded into
``exc`` = 0 and ``ecoul`` = 1/2 tr(D veff)), so that the driver's whole control flow --
localisation, subsystem energies, embedding potential, both projectors, environment deletion,
concentric localisation, Hamiltonian -- can run without PySCF.  The global object is the
oracle's numpy ToyUHF (it is an INPUT producer, outside the hot path); the embedded object is the
product's GpuUHF on whatever backend the test injects."""

import numpy as np

from oracle import synth
from oracle.pyscf_like import ToyMol, ToyUHF

from nbed_amd.scf import GpuUHF, Mole


class TaggedArray(np.ndarray):
    """ndarray carrying .ecoul / .exc like PySCF's tagged veff."""


class ToyKS(ToyUHF):
    """Global mean field whose "functional" is exact exchange: veff = J - K,
    ecoul = 1/2 tr(D_tot J), exc = -1/2 sum_x tr(D_x K_x).  With it HF-in-HF embedding is exact
    (e_rhf equals the global energy), the analogue of the reference's DFT-in-DFT check
    (tests/test_driver.py:83-88)."""

    xc = "exact-exchange"

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
        dm = self.make_rdm1() if dm is None else np.asarray(dm)
        dm3 = np.array((dm * 0.5, dm * 0.5)) if dm.ndim == 2 else dm
        vj, vk = self.get_jk(mol, dm3)
        v = (vj[0] + vj[1] - vk).view(TaggedArray)
        v.ecoul = 0.5 * float(np.einsum("ij,ji->", vj[0] + vj[1], dm3[0] + dm3[1]))
        v.exc = -0.5 * float(np.einsum("xij,xji->", vk, dm3))
        return v


class SyntheticProvider:
    def __init__(self, nao, nocc, n_act_aos, e_nuc=1.25, integrals=None, slices=None):
        self.nao, self.nocc, self.n_act_aos, self.e_nuc = nao, nocc, n_act_aos, e_nuc
        if integrals is None:
            self.S, self.h, self.eri = synth.overlap(nao), synth.hcore(nao), synth.eri_dense(nao)
        else:
            self.S, self.h, self.eri = integrals
        self.slices = slices if slices is not None else [[0, 1, 0, n_act_aos], [1, 2, n_act_aos, nao]]

    @classmethod
    def water_sto3g(cls, g):
        """Real molecule: the reference's water/STO-3G test system from tests/golden/water_sto3g.npz."""
        slices = [list(map(int, r)) for r in g["ao_slices"]]
        return cls(int(g["S"].shape[0]), tuple(int(x) for x in g["nelec"]), slices[0][3], float(g["e_nuc"]),
                   integrals=(g["S"], g["T"] + g["V"], g["eri"]), slices=slices)

    def build_mol(self, config):
        return Mole(self.nao, self.nocc, ao_slices=self.slices, e_nuc=self.e_nuc, atom=config.geometry,
                    basis=config.basis)

    def global_ks(self, config):
        mol = ToyMol(self.nao, self.nocc, ao_slices=self.slices, e_nuc=self.e_nuc)
        ks = ToyKS(mol, self.S, self.h, self.eri)
        ks.conv_tol, ks.max_cycle = 1e-11, 100
        ks.kernel()
        assert ks.converged
        return ks

    def local_hf(self, config, embedded_mol, backend=None):
        return GpuUHF(embedded_mol, self.S, self.h, self.eri, backend=backend)
