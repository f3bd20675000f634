"""Name-compatible stand-ins for the PySCF classes the reference type-checks against.

The reference selects energy formulas and restricted/unrestricted handling with
``isinstance(scf_method, (scf.rhf.RHF, scf.uhf.UHF, dft.rks.RKS, dft.uks.UKS))``
(nbed/scf/huzinaga_scf.py:176-187, nbed/ham_builder.py:43,277-283).  The GPU-backed SCF
objects of this package derive from the marker classes below; objects from a real PySCF
installation are recognised by duck typing (``is_ks`` / ``is_unrestricted``).
"""

from __future__ import annotations

import copy as _copy

import numpy as np


class StreamObject:
    """``pyscf.lib.StreamObject`` behaviours the path relies on."""

    verbose = 1

    def __call__(self, **kwargs):  # PySCF: __call__ = set, returns self
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    set = __call__

    def copy(self):
        """Shallow copy: instance-level monkey patches survive (driver.py:939)."""
        return _copy.copy(self)

    def run(self, *args, **kwargs):
        self.kernel(*args, **kwargs)
        return self


class RHF(StreamObject):
    """Marker: restricted Hartree-Fock."""


class UHF(StreamObject):
    """Marker: unrestricted Hartree-Fock."""


class RKS(StreamObject):
    """Marker: restricted Kohn-Sham."""


class UKS(StreamObject):
    """Marker: unrestricted Kohn-Sham."""


def is_ks(scf_method) -> bool:
    """Kohn-Sham object (ours or PySCF's: those carry an ``xc`` attribute)."""
    return isinstance(scf_method, (RKS, UKS)) or hasattr(scf_method, "xc")


def is_hf(scf_method) -> bool:
    if isinstance(scf_method, (RHF, UHF)):
        return True
    # a PySCF HF object: has the SCF protocol but no xc functional
    return (not hasattr(scf_method, "xc")) and hasattr(scf_method, "get_veff") and hasattr(scf_method, "get_occ")


def is_restricted(scf_method) -> bool:
    """ham_builder.py:43 -- isinstance(scf_method, (scf.rhf.RHF, dft.rks.RKS))."""
    if isinstance(scf_method, (RHF, RKS)):
        return True
    if isinstance(scf_method, (UHF, UKS)):
        return False
    mo = getattr(scf_method, "mo_coeff", None)
    return mo is not None and np.ndim(mo) == 2
