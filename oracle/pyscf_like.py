"""Restatement of the PySCF 2.9.0 behaviours the hot path relies on.

TEST INFRASTRUCTURE (see oracle/__init__.py).

PySCF is a third-party dependency of the reference (uv.lock:2571-2572) and is
not vendored in /root/reference nor installed here, so this module restates its
*published* algorithms in numpy:

* ``DIIS``      -- ``pyscf.lib.diis.DIIS`` as used at nbed/scf/huzinaga_scf.py:130,164
* ``ToyUHF``    -- the duck-typed SCF-object protocol of SURVEY.md section 8b
                   (``get_ovlp/get_hcore/get_veff/get_j/get_occ/make_rdm1/get_fock/
                   energy_elec/energy_tot/energy_nuc/eig/kernel/copy``) with the
                   J/K contraction done by einsum on a dense (pq|rs) array --
                   call sites nbed/scf/huzinaga_scf.py:156, nbed/driver.py:533
* ``ToyRHF``    -- the restricted flavour (nbed/scf/huzinaga_scf.py 2-D branch)
"""

from __future__ import annotations

import copy as _copy

import numpy as np
import scipy.linalg


class ToyMol:
    """The handful of ``gto.Mole`` attributes the path reads."""

    def __init__(self, nao, nelec, ao_slices=None, e_nuc=0.0, atom=None, basis="toy", charge=0):
        self.nao = int(nao)
        self.nelec = (int(nelec[0]), int(nelec[1]))
        self.nelectron = self.nelec[0] + self.nelec[1]
        self.spin = self.nelec[0] - self.nelec[1]
        self._ao_slices = ao_slices
        self._e_nuc = float(e_nuc)
        self.atom = atom
        self.basis = basis
        self.charge = charge

    def aoslice_by_atom(self):
        """(natm, 4) int array [shell0, shell1, ao0, ao1]."""
        if self._ao_slices is None:
            return np.array([[0, 1, 0, self.nao]])
        return np.asarray(self._ao_slices)

    def nao_nr(self):
        return self.nao

    def energy_nuc(self):
        return self._e_nuc


def get_jk(eri: np.ndarray, dm: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """J_pq = sum_rs (pq|rs) D_rs ; K_pr = sum_qs (pq|rs) D_qs, batched over dm[...]."""
    dm = np.asarray(dm)
    n = eri.shape[0]
    e2 = eri.reshape(n * n, n * n)
    dms = dm.reshape(-1, n, n)
    vj = (dms.reshape(-1, n * n) @ e2.T).reshape(-1, n, n)
    # K: (pq|rs) -> [p r][q s]
    ek = eri.transpose(0, 2, 1, 3).reshape(n * n, n * n)
    vk = (dms.reshape(-1, n * n) @ ek.T).reshape(-1, n, n)
    return vj.reshape(dm.shape), vk.reshape(dm.shape)


class DIIS:
    """``pyscf.lib.diis.DIIS`` (plain Pulay on the vector itself).

    space=6, min_space=1.  No error vector is pushed by the caller
    (nbed/scf/huzinaga_scf.py:164 calls ``adiis.update(fock)``), so the error
    is ``x - x_prev_returned``; the first ``update`` only stores ``x``.
    """

    def __init__(self, space: int = 6, min_space: int = 1):
        self.space = space
        self.min_space = min_space
        self._head = 0
        self._bookkeep: list[int] = []
        self._xprev = None
        self._x: dict[int, np.ndarray] = {}
        self._e: dict[int, np.ndarray] = {}
        self._H = None

    def push_vec(self, x):
        x = np.asarray(x).ravel()
        while len(self._bookkeep) >= self.space:
            self._bookkeep.pop(0)
        if self._xprev is None:
            self._xprev = x.copy()
        else:
            if self._head >= self.space:
                self._head = 0
            self._bookkeep.append(self._head)
            self._x[self._head] = x.copy()
            self._e[self._head] = x - self._xprev
            self._head += 1

    def get_num_vec(self):
        return len(self._bookkeep)

    def update(self, x):
        self.push_vec(x)
        nd = self.get_num_vec()
        if nd < self.min_space:
            return x
        dt = self._e[self._head - 1]
        if self._H is None:
            self._H = np.zeros((self.space + 1, self.space + 1))
            self._H[0, 1:] = self._H[1:, 0] = 1
        for i in range(nd):
            tmp = np.dot(dt, self._e[i])
            self._H[self._head, i + 1] = tmp
            self._H[i + 1, self._head] = tmp
        xnew = self.extrapolate(nd)
        self._xprev = xnew
        return xnew.reshape(np.shape(x))

    def extrapolate(self, nd):
        c = diis_coefficients(self._H[: nd + 1, : nd + 1])
        xnew = np.zeros_like(self._x[0])
        for i, ci in enumerate(c[1:]):
            xnew += self._x[i] * ci
        return xnew


def diis_coefficients(h: np.ndarray) -> np.ndarray:
    """Solve the Pulay system H c = (1,0,..); drop |w|<1e-14 modes if singular."""
    g = np.zeros(h.shape[0])
    g[0] = 1
    w, v = scipy.linalg.eigh(h)
    if np.any(abs(w) < 1e-14):
        idx = abs(w) > 1e-14
        return np.dot(v[:, idx] * (1.0 / w[idx]), np.dot(v[:, idx].T.conj(), g))
    return np.linalg.solve(h, g)


class CDIIS:
    """``pyscf.scf.diis.CDIIS``: error SDF-FDS in an orthonormal basis, space 8."""

    def __init__(self, space: int = 8):
        self.space = space
        self._vecs: list[np.ndarray] = []
        self._errs: list[np.ndarray] = []

    def update(self, s, d, f):
        if f.ndim == 3:
            errs = []
            for x in range(f.shape[0]):
                sdf = s @ d[x] @ f[x]
                errs.append(sdf.conj().T - sdf)
            err = np.concatenate([e.ravel() for e in errs])
        else:
            sdf = s @ d @ f
            err = (sdf.conj().T - sdf).ravel()
        self._vecs.append(f.ravel().copy())
        self._errs.append(err)
        if len(self._vecs) > self.space:
            self._vecs.pop(0)
            self._errs.pop(0)
        nd = len(self._vecs)
        h = np.zeros((nd + 1, nd + 1))
        h[0, 1:] = h[1:, 0] = 1
        for i in range(nd):
            for j in range(i + 1):
                h[i + 1, j + 1] = h[j + 1, i + 1] = np.dot(self._errs[i], self._errs[j])
        c = diis_coefficients(h)
        fnew = np.zeros_like(self._vecs[0])
        for ci, v in zip(c[1:], self._vecs):
            fnew += ci * v
        return fnew.reshape(f.shape)


class _SCFBase:
    """Common pieces of the duck-typed SCF object."""

    def __init__(self, mol: ToyMol, s, h, eri):
        self.mol = mol
        self._s = np.asarray(s)
        self._h = np.asarray(h)
        self._eri = eri
        self.mo_coeff = None
        self.mo_occ = None
        self.mo_energy = None
        self.e_tot = None
        self.converged = False
        self.max_cycle = 50
        self.conv_tol = 1e-9
        self.max_memory = 4000
        self.verbose = 1
        self.scf_summary = {}
        self.cycles = 0

    # PySCF's StreamObject.__call__ is ``set`` and returns self
    def __call__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        return self

    def copy(self):
        return _copy.copy(self)

    def run(self):
        self.kernel()
        return self

    def get_ovlp(self, mol=None):
        return self._s

    def get_hcore(self, mol=None):
        return self._h

    def energy_nuc(self):
        return self.mol.energy_nuc()

    @property
    def nelec(self):
        return self.mol.nelec


def eig_generalized(f, s):
    """``hf.eig``: scipy.linalg.eigh(F, S) + largest-|component|-positive sign rule."""
    e, c = scipy.linalg.eigh(f, s)
    idx = np.argmax(abs(c.real), axis=0)
    c[:, c[idx, np.arange(len(e))].real < 0] *= -1
    return e, c


class ToyUHF(_SCFBase):
    """Unrestricted SCF object over dense S, hcore, (pq|rs)."""

    def get_jk(self, mol=None, dm=None):
        if dm is None:
            dm = self.make_rdm1()
        return get_jk(self._eri, np.asarray(dm))

    def get_j(self, mol=None, dm=None):
        return self.get_jk(mol, dm)[0]

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
        if dm is None:
            dm = self.make_rdm1()
        dm = np.asarray(dm)
        if dm.ndim == 2:
            dm = np.array((dm * 0.5, dm * 0.5))
        vj, vk = get_jk(self._eri, dm)
        return vj[0] + vj[1] - vk

    def get_occ(self, mo_energy=None, mo_coeff=None):
        if mo_energy is None:
            mo_energy = self.mo_energy
        mo_energy = np.asarray(mo_energy)
        mo_occ = np.zeros_like(mo_energy)
        na, nb = self.mol.nelec
        mo_occ[0, np.argsort(mo_energy[0])[:na]] = 1
        mo_occ[1, np.argsort(mo_energy[1])[:nb]] = 1
        return mo_occ

    def make_rdm1(self, mo_coeff=None, mo_occ=None):
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        if mo_occ is None:
            mo_occ = self.mo_occ
        da = (mo_coeff[0] * mo_occ[0]) @ mo_coeff[0].conj().T
        db = (mo_coeff[1] * mo_occ[1]) @ mo_coeff[1].conj().T
        return np.array((da, db))

    def get_fock(self, h1e=None, s1e=None, vhf=None, dm=None):
        if h1e is None:
            h1e = self.get_hcore()
        if vhf is None:
            vhf = self.get_veff(self.mol, dm if dm is not None else self.make_rdm1())
        return h1e + vhf

    def eig(self, f, s):
        ea, ca = eig_generalized(f[0], s)
        eb, cb = eig_generalized(f[1], s)
        return np.array((ea, eb)), np.array((ca, cb))

    def energy_elec(self, dm=None, h1e=None, vhf=None):
        """``uhf.energy_elec`` for a 2-D hcore (the driver replaces it for 3-D)."""
        if dm is None:
            dm = self.make_rdm1()
        if h1e is None:
            h1e = self.get_hcore()
        if vhf is None:
            vhf = self.get_veff(self.mol, dm)
        if h1e.ndim == 2:
            h1e = (h1e, h1e)
        e1 = np.einsum("ij,ji->", h1e[0], dm[0]) + np.einsum("ij,ji->", h1e[1], dm[1])
        e_coul = 0.5 * (np.einsum("ij,ji->", vhf[0], dm[0]) + np.einsum("ij,ji->", vhf[1], dm[1]))
        self.scf_summary["e1"] = e1
        self.scf_summary["e2"] = e_coul
        return e1 + e_coul, e_coul

    def energy_tot(self, dm=None, h1e=None, vhf=None):
        return self.energy_elec(dm, h1e, vhf)[0] + self.energy_nuc()

    def get_init_guess(self):
        """Core-Hamiltonian guess (the reference uses 'minao', which needs a real basis)."""
        h = self.get_hcore()
        if h.ndim == 2:
            h = np.array((h, h))
        e, c = self.eig(h, self.get_ovlp())
        return self.make_rdm1(c, self.get_occ(e, c))

    def kernel(self, dm0=None):
        """``scf.hf.kernel``: CDIIS from cycle 1, conv_check."""
        s = self.get_ovlp()
        h1e = self.get_hcore()
        if h1e.ndim == 2:
            h1e = np.array((h1e, h1e))
        dm = self.get_init_guess() if dm0 is None else dm0
        vhf = self.get_veff(self.mol, dm)
        e_tot = self.energy_tot(dm, h1e, vhf)
        diis = CDIIS()
        conv_tol_grad = np.sqrt(self.conv_tol)
        self.converged = False
        mo_energy = mo_coeff = mo_occ = None
        for cycle in range(self.max_cycle):
            dm_last, last_hf_e = dm, e_tot
            fock = h1e + vhf
            if cycle >= 1:
                fock = diis.update(s, dm, fock)
            mo_energy, mo_coeff = self.eig(fock, s)
            mo_occ = self.get_occ(mo_energy, mo_coeff)
            dm = self.make_rdm1(mo_coeff, mo_occ)
            vhf = self.get_veff(self.mol, dm)
            e_tot = self.energy_tot(dm, h1e, vhf)
            fock = h1e + vhf
            gorb = self._grad(mo_coeff, mo_occ, fock)
            norm_gorb = np.linalg.norm(gorb) / np.sqrt(max(gorb.size, 1))
            self.cycles = cycle + 1
            if abs(e_tot - last_hf_e) < self.conv_tol and norm_gorb < conv_tol_grad:
                self.converged = True
                break
        if self.converged:
            # extra DIIS-free cycle ("conv_check")
            fock = h1e + vhf
            mo_energy, mo_coeff = self.eig(fock, s)
            mo_occ = self.get_occ(mo_energy, mo_coeff)
            dm = self.make_rdm1(mo_coeff, mo_occ)
            vhf = self.get_veff(self.mol, dm)
            e_tot = self.energy_tot(dm, h1e, vhf)
        self.mo_energy, self.mo_coeff, self.mo_occ = mo_energy, mo_coeff, mo_occ
        self.e_tot = e_tot
        return e_tot

    @staticmethod
    def _grad(mo_coeff, mo_occ, fock):
        gs = []
        for x in range(2):
            occ = mo_occ[x] > 0
            vir = ~occ
            gs.append((mo_coeff[x][:, vir].conj().T @ fock[x] @ mo_coeff[x][:, occ]).ravel())
        return np.hstack(gs)


class TaggedArray(np.ndarray):
    """ndarray carrying .ecoul / .exc like PySCF's tagged Kohn-Sham veff."""

    ecoul = exc = None


class ToyUKS(ToyUHF):
    """Unrestricted Kohn-Sham object whose functional is a fraction ``hyb`` of exact exchange:
    veff[x] = J - hyb K[x], ecoul = 1/2 tr(Dtot J), exc = -hyb/2 sum_x tr(D[x] K[x]) -- what
    tests/golden/make_golden.py hands to the reference's KS branch
    (nbed/scf/huzinaga_scf.py:36-62,176-180) in place of a B3LYP ``dft.UKS``."""

    hyb = 0.2
    _is_ks = True

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
        dm = self.make_rdm1() if dm is None else np.asarray(dm)
        if dm.ndim == 2:
            dm = np.array((dm * 0.5, dm * 0.5))
        vj, vk = get_jk(self._eri, dm)
        v = (vj[0] + vj[1] - self.hyb * vk).view(TaggedArray)
        v.ecoul = 0.5 * float(np.einsum("ij,ji->", vj[0] + vj[1], dm[0] + dm[1]))
        v.exc = -0.5 * self.hyb * float(np.einsum("xij,xji->", vk, dm))
        return v


class ToyRHF(_SCFBase):
    """Restricted SCF object (2-D arrays), for the 2-D branches of the path."""

    def get_jk(self, mol=None, dm=None):
        return get_jk(self._eri, np.asarray(dm))

    def get_j(self, mol=None, dm=None):
        return self.get_jk(mol, dm)[0]

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
        if dm is None:
            dm = self.make_rdm1()
        vj, vk = get_jk(self._eri, np.asarray(dm))
        return vj - 0.5 * vk

    def get_occ(self, mo_energy=None, mo_coeff=None):
        mo_energy = np.asarray(mo_energy)
        mo_occ = np.zeros_like(mo_energy)
        mo_occ[np.argsort(mo_energy)[: self.mol.nelectron // 2]] = 2
        return mo_occ

    def make_rdm1(self, mo_coeff=None, mo_occ=None):
        if mo_coeff is None:
            mo_coeff = self.mo_coeff
        if mo_occ is None:
            mo_occ = self.mo_occ
        return (mo_coeff * mo_occ) @ mo_coeff.conj().T

    def get_fock(self, h1e=None, s1e=None, vhf=None, dm=None):
        if h1e is None:
            h1e = self.get_hcore()
        if vhf is None:
            vhf = self.get_veff(self.mol, dm if dm is not None else self.make_rdm1())
        return h1e + vhf

    def energy_elec(self, dm=None, h1e=None, vhf=None):
        if dm is None:
            dm = self.make_rdm1()
        if h1e is None:
            h1e = self.get_hcore()
        if vhf is None:
            vhf = self.get_veff(self.mol, dm)
        e1 = np.einsum("ij,ji->", h1e, dm)
        e_coul = 0.5 * np.einsum("ij,ji->", vhf, dm)
        return e1 + e_coul, e_coul

    def energy_tot(self, dm=None, h1e=None, vhf=None):
        return self.energy_elec(dm, h1e, vhf)[0] + self.energy_nuc()

    def eig(self, f, s):
        return eig_generalized(f, s)
