"""Data returned by the occupied localizers (mirror of nbed/localizers/system.py:8-36)."""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ..backend import get_backend


def _projector(c: np.ndarray, backend=None) -> np.ndarray:
    """C C^T over the last two axes (batched over a leading spin axis), on the GPU."""
    be = backend if backend is not None else get_backend()
    c = np.asarray(c)
    if c.shape[-1] == 0:
        return np.zeros(c.shape[:-1] + (c.shape[-2],))
    c_d = be.asarray(c)
    return be.to_host(be.gemm(c_d, c_d, "N", "T"))


@dataclass
class LocalizedSystem:
    """Required data from a localized system.

    active_mo_inds / enviro_mo_inds: indices of the active / environment occupied MOs;
    c_active, c_enviro, c_loc_occ (, c_loc_virt): localized MO coefficient matrices
    (columns are MOs); dm_active, dm_enviro, dm_loc_occ: C C^T, derived in __post_init__.
    """

    active_mo_inds: np.ndarray
    enviro_mo_inds: np.ndarray
    c_active: np.ndarray
    c_enviro: np.ndarray
    c_loc_occ: np.ndarray
    c_loc_virt: np.ndarray | None = None
    dm_active: np.ndarray = field(init=False)
    dm_enviro: np.ndarray = field(init=False)
    dm_loc_occ: np.ndarray = field(init=False)
    backend: object = field(default=None, repr=False, compare=False)

    def __post_init__(self):
        self.dm_active = _projector(self.c_active, self.backend)
        self.dm_enviro = _projector(self.c_enviro, self.backend)
        self.dm_loc_occ = _projector(self.c_loc_occ, self.backend)
