"""Names the reference exports whose algorithms are outside the hot path.

PM / Boys / IBO localisation are iterative Jacobi-sweep schemes inside PySCF's ``lo`` module
(nbed/localizers/occupied/pyscf.py); no configuration of BASELINE.json uses them
(component 8).  The classes stay importable so that code written
against ``nbed.localizers`` loads, and fail loudly when instantiated.
"""

from __future__ import annotations

from .base import OccupiedLocalizer


class _Unsupported(OccupiedLocalizer):
    _name = "this"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            f"{self._name} localisation is not part of the MI355X hot path (SPADE is); "
            "use localization='spade'."
        )

    def _localize_spin(self, c_matrix, occupancy, n_mo_overwrite=None):  # pragma: no cover
        raise NotImplementedError


class PMLocalizer(_Unsupported):
    _name = "Pipek-Mezey"


class BOYSLocalizer(_Unsupported):
    _name = "Boys"


class IBOLocalizer(_Unsupported):
    _name = "IBO"
