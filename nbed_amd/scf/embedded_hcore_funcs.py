"""Spin-aware electronic energy for a 3-D (per-spin) core Hamiltonian.

Mirror of nbed/scf/embedded_hcore_funcs.py:11-46 (``energy_elec``); the traces run on the
GPU (``nbx_trace_prod``).  ``_absorb_h1e`` of the reference (:49-83) is dead code (never
called) and is not reproduced.
"""

from __future__ import annotations

import numpy as np

from ..backend import get_backend


def energy_elec(mf, dm=None, h1e=None, vhf=None):
    """Electronic energy of unrestricted Hartree-Fock; updates ``mf.scf_summary``.

    Returns (e_elec, e_coul) with e1 = sum_x tr(h[x] D[x]), e_coul = 1/2 sum_x tr(vhf[x] D[x]).
    """
    be = getattr(mf, "be", None) or get_backend()
    if dm is None:
        dm = mf.make_rdm1()
    if h1e is None:
        h1e = mf.get_hcore()
    dm = np.asarray(dm)
    if dm.ndim == 2:
        dm = np.array((dm * 0.5, dm * 0.5))
    if vhf is None:
        vhf = mf.get_veff(mf.mol, dm)
    h1e = np.asarray(h1e)
    if h1e.ndim != 3:
        raise IndexError("energy_elec expects a per-spin (2,N,N) core Hamiltonian")
    dm_d = be.asarray(dm)
    e1 = float(be.trace_prod(be.asarray(h1e), dm_d).sum())
    e_coul = float(be.trace_prod(be.asarray(np.asarray(vhf)), dm_d).sum()) * 0.5
    e_elec = e1 + e_coul
    mf.scf_summary["e1"] = e1
    mf.scf_summary["e2"] = e_coul
    return e_elec, e_coul
