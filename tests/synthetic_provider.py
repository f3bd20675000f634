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
    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
        dm = self.make_rdm1() if dm is None else np.asarray(dm)
        dm3 = np.array((dm * 0.5, dm * 0.5)) if dm.ndim == 2 else dm
        v = super().get_veff(mol, dm3)
        out = v.view(TaggedArray)
        out.ecoul = 0.5 * float(np.einsum("xij,xji->", v, dm3))
        out.exc = 0.0
        return out


class SyntheticProvider:
    def __init__(self, nao, nocc, n_act_aos, e_nuc=1.25):
        self.nao, self.nocc, self.n_act_aos, self.e_nuc = nao, nocc, n_act_aos, e_nuc
        self.S, self.h = synth.overlap(nao), synth.hcore(nao)
        self.eri = synth.eri_dense(nao)
        self.slices = [[0, 1, 0, n_act_aos], [1, 2, n_act_aos, nao]]

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
