"""Oracle: SPADE occupied localisation and concentric virtual localisation.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
nbed/localizers/system.py:8-36, nbed/localizers/occupied/spade.py:57-147,
nbed/localizers/occupied/base.py:64-140 and
nbed/localizers/virtual/concentric.py:123-262.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .huzinaga import sym_power


@dataclass
class LocalizedSystem:
    """nbed/localizers/system.py:8-36."""

    active_mo_inds: np.ndarray
    enviro_mo_inds: np.ndarray
    c_active: np.ndarray
    c_enviro: np.ndarray
    c_loc_occ: np.ndarray
    c_loc_virt: np.ndarray | None = None
    dm_active: np.ndarray = field(init=False)
    dm_enviro: np.ndarray = field(init=False)
    dm_loc_occ: np.ndarray = field(init=False)

    def __post_init__(self):
        self.dm_active = self.c_active @ self.c_active.swapaxes(-1, -2)
        self.dm_enviro = self.c_enviro @ self.c_enviro.swapaxes(-1, -2)
        self.dm_loc_occ = self.c_loc_occ @ self.c_loc_occ.swapaxes(-1, -2)


def spade_partition(sigma: np.ndarray, n_mo_overwrite=None) -> int:
    """Number of active MOs from the SPADE singular values (spade.py:105-121)."""
    if len(sigma) == 1:
        return 1
    if n_mo_overwrite is not None and len(sigma) >= n_mo_overwrite:
        return int(n_mo_overwrite)
    value_diffs = sigma[:-1] - sigma[1:]
    if np.allclose(value_diffs, [0] * len(value_diffs)):
        return len(sigma)
    return int(np.argmax(value_diffs) + 1)


def spade_localize_spin(c_matrix, occupancy, s_mat, n_act_aos, n_mo_overwrite=None):
    """spade.py:57-147 for one spin; returns (LocalizedSystem, sigma)."""
    n_occ = np.count_nonzero(occupancy)
    occupied = c_matrix[:, :n_occ]
    rotated = sym_power(s_mat, 0.5) @ occupied
    _, sigma, right_vectors = np.linalg.svd(rotated[:n_act_aos, :])
    n_act = spade_partition(sigma, n_mo_overwrite)
    n_env = n_occ - n_act
    active_inds = np.arange(n_act)
    enviro_inds = np.arange(n_act, n_act + n_env)
    c_active = occupied @ right_vectors.T[:, :n_act]
    c_enviro = occupied @ right_vectors.T[:, n_act:]
    c_loc_occ = occupied @ right_vectors.T
    return LocalizedSystem(active_inds, enviro_inds, c_active, c_enviro, c_loc_occ), sigma


def spade_localize(mo_coeff, mo_occ, s_mat, n_act_aos, n_mo_overwrite=(None, None)):
    """OccupiedLocalizer.localize() with the SPADE spin kernel (base.py:64-140).

    Returns (LocalizedSystem, enviro_selection_condition) where the latter is
    the (sigma_alpha, sigma_beta) pair stored at spade.py:137-143.
    """
    mo_coeff = np.asarray(mo_coeff)
    if mo_coeff.ndim == 2:
        ls, sig = spade_localize_spin(mo_coeff, mo_occ, s_mat, n_act_aos, n_mo_overwrite[0])
        ls.dm_active *= 2.0
        ls.dm_enviro *= 2.0
        return ls, (sig, np.zeros(len(sig)))
    alpha, sa = spade_localize_spin(mo_coeff[0], mo_occ[0], s_mat, n_act_aos, n_mo_overwrite[0])
    beta, sb = spade_localize_spin(mo_coeff[1], mo_occ[1], s_mat, n_act_aos, n_mo_overwrite[1])
    cond = (sa, sb)
    ls = LocalizedSystem(
        np.array([alpha.active_mo_inds, beta.active_mo_inds]),
        np.array([alpha.enviro_mo_inds, beta.enviro_mo_inds]),
        np.array([alpha.c_active, beta.c_active]),
        np.array([alpha.c_enviro, beta.c_enviro]),
        np.array([alpha.c_loc_occ, beta.c_loc_occ]),
    )
    if set(alpha.active_mo_inds) != set(beta.active_mo_inds) or set(alpha.enviro_mo_inds) != set(
        beta.enviro_mo_inds
    ):
        occ_sum = np.sum(mo_occ, axis=0)
        a2, s2a = spade_localize_spin(mo_coeff[0], occ_sum, s_mat, n_act_aos, n_mo_overwrite[0])
        b2, s2b = spade_localize_spin(mo_coeff[1], occ_sum, s_mat, n_act_aos, n_mo_overwrite[1])
        # the reference keeps overwriting slot [1] of the stored condition
        cond = (sa, s2b)
        ls = LocalizedSystem(
            np.array([alpha.active_mo_inds, beta.active_mo_inds]),
            np.array([alpha.enviro_mo_inds, beta.enviro_mo_inds]),
            np.array([a2.c_active, b2.c_active]),
            np.array([a2.c_enviro, b2.c_enviro]),
            np.array([a2.c_loc_occ, b2.c_loc_occ]),
        )
    return ls, cond


def concentric_localize_spin(occ, mo_coeff, fock, projected_overlap, overlap_two_basis,
                             n_act_proj_aos, max_shells):
    """concentric.py:123-262.  Returns (mo_coeff, shells, singular_values)."""
    effective_virt = mo_coeff[:, occ == 0]
    left = np.linalg.inv(projected_overlap) @ overlap_two_basis @ effective_virt
    _, sigma, right_vectors = np.linalg.svd(
        np.swapaxes(left, -1, -2) @ overlap_two_basis @ effective_virt
    )
    singular_values = [sigma]
    c_total = mo_coeff[:, occ > 0]
    shell_size = np.sum(sigma[:n_act_proj_aos] >= 1e-15)
    right_vectors = np.swapaxes(right_vectors, -1, -2)
    v_span, v_ker = np.split(right_vectors, [shell_size], axis=-1)
    c_ispan = effective_virt @ v_span
    c_iker = effective_virt @ v_ker
    c_total = np.concatenate((c_total, c_ispan), axis=-1)
    shells = [c_total.shape[-1]]

    if v_ker.shape[-1] == 0:
        pass
    elif v_ker.shape[-1] == 1:
        c_total = np.concatenate((c_total, c_iker), axis=-1)
        shells.append(c_total.shape[-1])
    else:
        for ishell in range(0, max_shells):
            _, sigma, right_vectors = np.linalg.svd(np.swapaxes(c_total, -1, -2) @ fock @ c_iker)
            singular_values.append(sigma)
            shell_size = np.sum(sigma[:n_act_proj_aos] >= 1e-15)
            if shell_size == 0:
                c_total = np.concatenate((c_total, c_iker), axis=-1)
                break
            right_vectors = np.swapaxes(right_vectors, -1, -2)
            v_span, v_ker = np.split(right_vectors, [shell_size], axis=-1)
            c_ispan = c_iker @ v_span
            c_total = np.concatenate((c_total, c_ispan), axis=-1)
            shells.append(c_total.shape[-1])
            if v_ker.shape[-1] > 1:
                c_iker = c_iker @ v_ker
            elif v_ker.shape[-1] == 1:
                c_iker = c_iker @ v_ker
                c_total = np.concatenate((c_total, c_iker), axis=-1)
                shells.append(c_total.shape[-1])
                break
            else:
                break
            if ishell >= max_shells:  # unreachable in the reference too (:249)
                c_total = np.concatenate((c_total, c_iker), axis=-1)
                shells.append(c_total.shape[-1])
                break
    return c_total, shells, singular_values
