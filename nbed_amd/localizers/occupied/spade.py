"""SPADE localizer on the GPU (mirror of nbed/localizers/occupied/spade.py:57-147).

Per spin: rotate the occupied orbitals into the Loewdin basis, M = (S^1/2 C_occ)[:n_act_aos],
take the SVD of M (``nbx_svd_right``: one-sided Jacobi, sigma and the full V), cut at the
largest gap of the singular values (:105-121), and rotate C_occ by V (:132-134).
"""

from __future__ import annotations

import logging

import numpy as np

from ..system import LocalizedSystem
from .base import OccupiedLocalizer

logger = logging.getLogger(__name__)


class SPADELocalizer(OccupiedLocalizer):
    """Localise occupied MOs with SPADE; returns active and environment systems."""

    def __init__(self, global_scf, n_active_atoms: int, max_shells: int = 4,
                 n_mo_overwrite: tuple[int | None, int | None] | None = None, backend=None):
        self.max_shells = max_shells
        self.shells = None
        self.singular_values = None
        self.enviro_selection_condition = None
        self._s_half = None
        super().__init__(global_scf, n_active_atoms, n_mo_overwrite, backend=backend)

    def _localize_spin(self, c_matrix: np.ndarray, occupancy: np.ndarray,
                       n_mo_overwrite: int | None = None) -> LocalizedSystem:
        be = self._be
        c_matrix = np.asarray(c_matrix)
        n_occupied_orbitals = int(np.count_nonzero(occupancy))
        occupied_orbitals = np.ascontiguousarray(c_matrix[:, :n_occupied_orbitals])
        n_act_aos = int(self._global_scf.mol.aoslice_by_atom()[self._n_active_atoms - 1][-1])

        if self._s_half is None:  # the reference recomputes S^1/2 per spin (:99); same matrix
            s_h = np.asarray(self._global_scf.get_ovlp())
            fast = getattr(be, "sym_pow_fast", None)
            self._s_half = fast(be.asarray(s_h), 0.5, s_h) if fast is not None else be.sym_pow(be.asarray(s_h), 0.5)
        occ_d = be.asarray(occupied_orbitals)
        rotated = be.gemm(self._s_half, occ_d)
        sigma_d, vt_d = be.svd_right(rotated[:n_act_aos, :].contiguous())
        sigma = be.to_host(sigma_d)

        if len(sigma) == 1:
            n_act_mos = 1
        elif n_mo_overwrite is not None and len(sigma) >= n_mo_overwrite:
            n_act_mos = n_mo_overwrite
        else:
            value_diffs = sigma[:-1] - sigma[1:]
            if np.allclose(value_diffs, [0] * len(value_diffs)):
                n_act_mos = len(sigma)
            else:
                n_act_mos = int(np.argmax(value_diffs) + 1)
        n_env_mos = n_occupied_orbitals - n_act_mos

        active_mo_inds = np.arange(n_act_mos)
        enviro_mo_inds = np.arange(n_act_mos, n_act_mos + n_env_mos)

        c_loc_occ = be.to_host(be.gemm(occ_d, vt_d, "N", "T"))  # C_occ V
        c_active = np.ascontiguousarray(c_loc_occ[:, :n_act_mos])
        c_enviro = np.ascontiguousarray(c_loc_occ[:, n_act_mos:])

        if self.enviro_selection_condition is None:
            self.enviro_selection_condition = (sigma, np.zeros(len(sigma)))
        else:
            self.enviro_selection_condition = (self.enviro_selection_condition[0], sigma)

        return LocalizedSystem(active_mo_inds, enviro_mo_inds, c_active, c_enviro, c_loc_occ, backend=be)
