"""Oracle: Huzinaga-projected SCF loop and spin-aware energy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
nbed/scf/huzinaga_scf.py:65-206 and nbed/scf/embedded_hcore_funcs.py:11-46.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg

from .pyscf_like import DIIS, ToyRHF, ToyUHF


def sym_power(s: np.ndarray, p: float) -> np.ndarray:
    """S**p for SPD S through the spectral decomposition.

    The reference uses scipy.linalg.fractional_matrix_power (Schur based,
    nbed/scf/huzinaga_scf.py:128, nbed/localizers/occupied/spade.py:99); for a
    symmetric positive definite S both give the principal power; they agree to
    rounding (checked in tests/test_oracle_golden.py).
    """
    w, u = np.linalg.eigh(s)
    return (u * w**p) @ u.T


def get_huzinaga_operator(fock, dm_occ_s, dm_virt_s):
    """nbed/scf/huzinaga_scf.py:65-90."""
    fds_occ = np.einsum("...ij,...jk->...ik", fock, dm_occ_s)
    huz_occ = fds_occ + np.swapaxes(fds_occ, -1, -2)
    huz_occ *= (-0.5) if fds_occ.ndim == 2 else (-1.0)

    fds_virt = np.einsum("...ij,...jk->...ik", fock, dm_virt_s)
    huz_virt = (
        fds_virt
        + np.swapaxes(fds_virt, -1, -2)
        - 2 * np.einsum("...ij,...jk->...ik", np.swapaxes(dm_virt_s, -1, -2), fds_virt)
    )
    huz_virt *= (-0.5) if fds_virt.ndim == 2 else (-1.0)
    return huz_occ + huz_virt


def calculate_ks_energy(scf_method, embedding_potential, density_matrix, huzinaga_op_occ):
    """nbed/scf/huzinaga_scf.py:36-62: E_coul + E_xc of the NEW density (a second get_veff) plus
    tr[D (hcore + Hz + V_emb)] -- one value per spin for 3-D input."""
    vhf_updated = scf_method.get_veff(dm=density_matrix)
    energy = vhf_updated.ecoul + vhf_updated.exc
    energy = energy + np.einsum(
        "...ij,...ji->...", density_matrix, scf_method.get_hcore() + huzinaga_op_occ + embedding_potential
    )
    return energy


def huzinaga_scf(
    scf_method,
    embedding_potential,
    dm_environment_occupied,
    dm_environment_virtual=None,
    dm_conv_tol=1e-6,
    dm_initial_guess=None,
    use_DIIS=True,
    exact_power=False,
    history=None,
):
    """nbed/scf/huzinaga_scf.py:93-206 (HF branch ``:181-185``; KS branch ``:176-180``).

    ``history`` (optional list) receives (energy, dm_diff) per cycle.
    ``exact_power`` switches S^-1/2 to scipy's fractional_matrix_power.
    """
    s_mat = scf_method.get_ovlp()
    if exact_power:
        s_neg_half = scipy.linalg.fractional_matrix_power(s_mat, -0.5)
    else:
        s_neg_half = sym_power(s_mat, -0.5)
    adiis = DIIS() if use_DIIS else None

    dm_occ_s = np.einsum("...ij,jk->...ik", dm_environment_occupied, s_mat)
    if dm_environment_virtual is not None:
        dm_virt_s = np.einsum("...ij,jk->...ik", dm_environment_virtual, s_mat)
    else:
        dm_virt_s = np.zeros(dm_occ_s.shape)

    if dm_initial_guess is None:
        fock = scf_method.get_hcore() + embedding_potential
        fock = fock + get_huzinaga_operator(fock, dm_occ_s, dm_virt_s)
        fock_ortho = s_neg_half @ fock @ s_neg_half
        mo_energy, mo_coeff_ortho = np.linalg.eigh(fock_ortho)
        mo_coeff_std = s_neg_half @ mo_coeff_ortho
        mo_occ = scf_method.get_occ(mo_energy, mo_coeff_std)
        dm_initial_guess = scf_method.make_rdm1(mo_coeff=mo_coeff_std, mo_occ=mo_occ)

    density_matrix = dm_initial_guess
    conv_flag = False
    scf_energy_prev = 0

    for i in range(scf_method.max_cycle):
        vhf = scf_method.get_veff(dm=density_matrix)
        fock = scf_method.get_hcore() + embedding_potential + vhf
        huzinaga_op = get_huzinaga_operator(fock, dm_occ_s, dm_virt_s)
        fock = fock + huzinaga_op
        if use_DIIS and (i > 1):
            fock = adiis.update(fock)
        fock_ortho = s_neg_half @ fock @ s_neg_half
        mo_energy, mo_coeff_ortho = np.linalg.eigh(fock_ortho)
        mo_coeff_std = s_neg_half @ mo_coeff_ortho
        mo_occ = scf_method.get_occ(mo_energy, mo_coeff_std)
        dm_mat_old = density_matrix
        density_matrix = scf_method.make_rdm1(mo_coeff=mo_coeff_std, mo_occ=mo_occ)

        if getattr(scf_method, "_is_ks", False):  # isinstance(..., (RKS, UKS)) is tested first (:176)
            scf_energy = calculate_ks_energy(scf_method, embedding_potential, density_matrix, huzinaga_op)
        elif isinstance(scf_method, (ToyRHF, ToyUHF)) or getattr(scf_method, "_is_hf", False):
            hamiltonian = scf_method.get_hcore() + embedding_potential + 0.5 * vhf + huzinaga_op
            scf_energy = np.einsum("...ij,...ji->...", hamiltonian, density_matrix)
        else:
            raise TypeError("Cannot run Huzinaga SCF with type %s" % type(scf_method))

        run_diff = np.max(np.abs(scf_energy - scf_energy_prev))
        norm_dm_diff = np.max(np.linalg.norm(density_matrix - dm_mat_old, axis=(-2, -1)))
        if history is not None:
            history.append((np.array(scf_energy, copy=True), float(norm_dm_diff)))
        if (run_diff < scf_method.conv_tol) and (norm_dm_diff < dm_conv_tol):
            conv_flag = True
            break
        scf_energy_prev = scf_energy

    return mo_coeff_std, mo_energy, density_matrix, huzinaga_op, conv_flag


def energy_elec(mf, dm=None, h1e=None, vhf=None):
    """nbed/scf/embedded_hcore_funcs.py:11-46 (3-D hcore aware)."""
    if dm is None:
        dm = mf.make_rdm1()
    if h1e is None:
        h1e = mf.get_hcore()
    if isinstance(dm, np.ndarray) and dm.ndim == 2:
        dm = np.array((dm * 0.5, dm * 0.5))
    if vhf is None:
        vhf = mf.get_veff(mf.mol, dm)
    e1 = np.einsum("ij,ji->", h1e[0], dm[0])
    e1 += np.einsum("ij,ji->", h1e[1], dm[1])
    e_coul = (np.einsum("ij,ji->", vhf[0], dm[0]) + np.einsum("ij,ji->", vhf[1], dm[1])) * 0.5
    e_elec = (e1 + e_coul).real
    mf.scf_summary["e1"] = e1.real
    mf.scf_summary["e2"] = e_coul.real
    return e_elec, e_coul
