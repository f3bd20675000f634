"""Base class of the virtual-orbital localizers (mirror of nbed/localizers/virtual/base.py:8-36)."""

from abc import ABC, abstractmethod


class VirtualLocalizer(ABC):
    """Plugin surface: subclasses implement ``localize_virtual()`` returning the SCF object."""

    def __init__(self, n_active_atoms: int):
        self._n_active_atoms = n_active_atoms

    @abstractmethod
    def localize_virtual(self):
        """Localize virtual orbitals; returns the (modified) SCF object."""
