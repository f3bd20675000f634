"""Base class of the occupied-orbital localizers -- the Localizer plugin surface.

Mirror of nbed/localizers/occupied/base.py: subclasses implement
``_localize_spin(c_matrix, occupancy, n_mo_overwrite) -> LocalizedSystem`` (:142-159);
``localize()`` runs it for alpha and beta (:64-140).  Reference behaviour kept as is:

* restricted input: the derived density matrices are doubled (:84-85);
* the alpha/beta index arrays are packed with ``np.array([...])`` (:98-100), so occupations
  that differ between spins (open shells) raise ValueError with numpy >= 1.24 exactly as the
  reference does -- the consistency re-run (:107-130) is therefore only reachable when both
  arrays have equal lengths.
"""

from __future__ import annotations

import logging
from abc import ABC, abstractmethod

import numpy as np

from ...backend import get_backend
from ...exceptions import NbedLocalizerError
from ..system import LocalizedSystem

logger = logging.getLogger(__name__)


class OccupiedLocalizer(ABC):
    """Localise occupied MOs into an active and an environment subsystem."""

    def __init__(self, global_scf, n_active_atoms: int,
                 n_mo_overwrite: tuple[int | None, int | None] | None = None, backend=None):
        if global_scf.mo_coeff is None:
            logger.debug("SCF method not initialised, running now...")
            global_scf.run()
        self.n_mo_overwrite = (None, None) if n_mo_overwrite is None else n_mo_overwrite
        self._global_scf = global_scf
        self._n_active_atoms = n_active_atoms
        self._be = backend if backend is not None else (getattr(global_scf, "be", None) or get_backend())
        self.spinless = np.asarray(global_scf.mo_coeff).ndim == 2

    def localize(self) -> LocalizedSystem:
        scf = self._global_scf
        if self.spinless:
            localized_system = self._localize_spin(scf.mo_coeff, scf.mo_occ, self.n_mo_overwrite[0])
            localized_system.dm_active *= 2.0
            localized_system.dm_enviro *= 2.0
        else:
            alpha = self._localize_spin(scf.mo_coeff[0], scf.mo_occ[0], self.n_mo_overwrite[0])
            beta = self._localize_spin(scf.mo_coeff[1], scf.mo_occ[1], self.n_mo_overwrite[1])
            localized_system = LocalizedSystem(
                np.array([alpha.active_mo_inds, beta.active_mo_inds]),
                np.array([alpha.enviro_mo_inds, beta.enviro_mo_inds]),
                np.array([alpha.c_active, beta.c_active]),
                np.array([alpha.c_enviro, beta.c_enviro]),
                np.array([alpha.c_loc_occ, beta.c_loc_occ]),
                backend=self._be,
            )
            if set(alpha.active_mo_inds) != set(beta.active_mo_inds) or set(alpha.enviro_mo_inds) != set(
                beta.enviro_mo_inds
            ):
                logger.debug("Recalculating occupied embedded C matrices to enforce equal number between spins.")
                mo_occ_sum = np.sum(scf.mo_occ, axis=0)
                alpha_consistent = self._localize_spin(scf.mo_coeff[0], mo_occ_sum, self.n_mo_overwrite[0])
                consistent = self._localize_spin(scf.mo_coeff[1], mo_occ_sum, self.n_mo_overwrite[1])
                localized_system = LocalizedSystem(
                    np.array([alpha.active_mo_inds, beta.active_mo_inds]),
                    np.array([alpha.enviro_mo_inds, beta.enviro_mo_inds]),
                    np.array([alpha_consistent.c_active, consistent.c_active]),
                    np.array([alpha_consistent.c_enviro, consistent.c_enviro]),
                    np.array([alpha_consistent.c_loc_occ, consistent.c_loc_occ]),
                    backend=self._be,
                )
        return localized_system

    @abstractmethod
    def _localize_spin(self, c_matrix: np.ndarray, occupancy: np.ndarray,
                       n_mo_overwrite: int | None = None) -> LocalizedSystem:
        """Localize the orbitals of one spin."""


def check_values(localized_system: LocalizedSystem, global_scf) -> None:
    """Sanity checks of a localisation (nbed/localizers/occupied/base.py:162-248):
    equal alpha/beta orbital counts, D_act + D_env = D_loc, electron number conserved."""
    warn_flag = False
    if np.asarray(localized_system.active_mo_inds).ndim == 2:
        if (localized_system.active_mo_inds[0].shape != localized_system.active_mo_inds[1].shape
                or localized_system.enviro_mo_inds[0].shape != localized_system.enviro_mo_inds[1].shape):
            logger.error("Number of alpha and beta orbitals do not match.")
            warn_flag = True

    c = np.asarray(localized_system.c_loc_occ)
    dm_full = c @ np.swapaxes(c.conj(), -1, -2)
    dm_sum = localized_system.dm_active + localized_system.dm_enviro
    density_match = np.allclose(2 * dm_full, dm_sum) if c.ndim == 2 else np.allclose(dm_full, dm_sum)
    if not density_match:
        logger.error("Density matrix partition does not sum to total.")
        warn_flag = True

    s = global_scf.get_ovlp()
    if localized_system.dm_active.ndim == 2:
        n_act = np.trace(localized_system.dm_active @ s)
        n_env = np.trace(localized_system.dm_enviro @ s)
    else:
        n_act = np.trace(localized_system.dm_active[0] @ s) + np.trace(localized_system.dm_active[1] @ s)
        n_env = np.trace(localized_system.dm_enviro[0] @ s) + np.trace(localized_system.dm_enviro[1] @ s)
    if not np.isclose(n_act + n_env, global_scf.mol.nelectron):
        logger.error("Number of electrons in localized orbitals is not consistent.")
        warn_flag = True

    if warn_flag:
        raise NbedLocalizerError("Localizer sense check failed.\n")
