"""Concentric localisation of the virtual orbitals on the GPU.

Mirror of nbed/localizers/virtual/concentric.py (Claudino & Mayhall, JCTC 15, 6085 (2019)):
shell 0 from the SVD of (S_AA^-1 S_AB C_virt)^T S_AB C_virt (:146-153), then up to
``max_shells`` SVDs of C_tot^T F C_ker (:198-254), splitting the right vectors into span and
kernel with the reference's ``sigma >= 1e-15`` test (:164,211).  The projected basis equals
the working basis (as in the reference, :75), so S_AA and S_AB are blocks of the AO overlap.
GEMMs -> nbx_gemm, S_AA^-1 -> nbx_sym_pow(-1), SVDs -> nbx_svd_right (one-sided Jacobi keeps
tiny singular values meaningful for the 1e-15 test).
"""

from __future__ import annotations

import logging

import numpy as np

from ...backend import get_backend
from .base import VirtualLocalizer

logger = logging.getLogger(__name__)


class ConcentricLocalizer(VirtualLocalizer):
    """Localize virtual orbitals shell by shell around the active atoms."""

    def __init__(self, embedded_scf, n_active_atoms: int, max_shells: int = 4, backend=None):
        super().__init__(n_active_atoms)
        self.embedded_scf = embedded_scf
        self.max_shells = max_shells
        self.projected_overlap = None
        self.overlap_two_basis = None
        self.n_act_proj_aos = None
        self.shells = None
        self.singular_values = None
        self._be = backend if backend is not None else (getattr(embedded_scf, "be", None) or get_backend())

    def localize_virtual(self):
        scf = self.embedded_scf
        n_act_proj_aos = int(scf.mol.aoslice_by_atom()[self._n_active_atoms - 1][-1])
        s = np.asarray(scf.get_ovlp())
        self.projected_overlap = s[:n_act_proj_aos, :n_act_proj_aos]
        self.overlap_two_basis = s[:n_act_proj_aos, :]
        self.n_act_proj_aos = n_act_proj_aos

        mo_coeff = np.asarray(scf.mo_coeff)
        if mo_coeff.ndim == 2:
            out = self._localize_virtual_spin(np.asarray(scf.mo_occ), mo_coeff, np.asarray(scf.get_fock()))
            scf.mo_coeff = out[0]
            self.shells = out[1]
            self.singular_values = out[2]
        else:
            a = self._localize_virtual_spin(np.asarray(scf.mo_occ[0]), mo_coeff[0], np.asarray(scf.get_fock())[0])
            b = self._localize_virtual_spin(np.asarray(scf.mo_occ[1]), mo_coeff[1], np.asarray(scf.get_fock())[1])
            scf.mo_coeff = np.array([a[0], b[0]])
            self.shells = (a[1], b[1])
            self.singular_values = (a[2], b[2])
        return scf

    def _svd(self, mat_d):
        s_d, vt_d = self._be.svd_right(mat_d)
        return self._be.to_host(s_d), vt_d

    def _localize_virtual_spin(self, occ: np.ndarray, mo_coeff: np.ndarray, fock_operator: np.ndarray):
        """One spin; returns (mo_coeff, shells, singular_values) (concentric.py:123-262)."""
        be = self._be
        occ = np.asarray(occ)
        effective_virt = np.ascontiguousarray(mo_coeff[:, occ == 0])
        c_total = np.ascontiguousarray(mo_coeff[:, occ > 0])
        nvirt = effective_virt.shape[1]

        s_ab = be.asarray(np.ascontiguousarray(self.overlap_two_basis))
        s_aa_inv = be.sym_pow(be.asarray(np.ascontiguousarray(self.projected_overlap)), -1.0)
        ev_d = be.asarray(effective_virt)
        sab_c = be.gemm(s_ab, ev_d)                       # S_AB C_virt          (n_act x n_virt)
        left = be.gemm(s_aa_inv, sab_c)                   # S_AA^-1 S_AB C_virt
        sigma, vt_d = self._svd(be.gemm(left, sab_c, "T", "N"))
        singular_values = [sigma]

        shell_size = int(np.sum(sigma[: self.n_act_proj_aos] >= 1e-15))
        v_d = be.transpose(vt_d)                          # right vectors as columns
        rotated = be.to_host(be.gemm(ev_d, v_d))          # C_virt V
        c_ispan, c_iker = rotated[:, :shell_size], rotated[:, shell_size:]
        c_total = np.concatenate((c_total, c_ispan), axis=-1)
        shells = [c_total.shape[-1]]

        n_ker = nvirt - shell_size
        if n_ker == 0:
            logger.debug("No kernel for 0th shell, cannot perform CL.")
        elif n_ker == 1:
            c_total = np.concatenate((c_total, c_iker), axis=-1)
            shells.append(c_total.shape[-1])
        else:
            fock_d = be.asarray(fock_operator)
            for ishell in range(0, self.max_shells):
                ct_d = be.asarray(np.ascontiguousarray(c_total))
                ck_d = be.asarray(np.ascontiguousarray(c_iker))
                sigma, vt_d = self._svd(be.gemm(be.gemm(ct_d, fock_d, "T", "N"), ck_d))
                singular_values.append(sigma)
                shell_size = int(np.sum(sigma[: self.n_act_proj_aos] >= 1e-15))
                if shell_size == 0:
                    c_total = np.concatenate((c_total, c_iker), axis=-1)
                    break
                rotated = be.to_host(be.gemm(ck_d, be.transpose(vt_d)))  # C_ker V
                c_ispan, c_new_ker = rotated[:, :shell_size], rotated[:, shell_size:]
                c_total = np.concatenate((c_total, c_ispan), axis=-1)
                shells.append(c_total.shape[-1])
                n_ker = c_new_ker.shape[-1]
                if n_ker > 1:
                    c_iker = c_new_ker
                elif n_ker == 1:
                    c_iker = c_new_ker
                    c_total = np.concatenate((c_total, c_iker), axis=-1)
                    shells.append(c_total.shape[-1])
                    break
                else:
                    break
                if ishell >= self.max_shells:  # kept from the reference (:249); never true
                    c_total = np.concatenate((c_total, c_iker), axis=-1)
                    shells.append(c_total.shape[-1])
                    break
        return c_total, shells, singular_values
