"""Localizer classes (same exports as nbed/localizers/__init__.py:10-19)."""

from .ace import ACELocalizer
from .occupied.base import OccupiedLocalizer
from .occupied.spade import SPADELocalizer
from .occupied.unsupported import BOYSLocalizer, IBOLocalizer, PMLocalizer
from .system import LocalizedSystem
from .virtual.base import VirtualLocalizer
from .virtual.concentric import ConcentricLocalizer
from .virtual.unsupported import PAOLocalizer

__all__ = [
    "ACELocalizer",
    "BOYSLocalizer",
    "IBOLocalizer",
    "PMLocalizer",
    "SPADELocalizer",
    "ConcentricLocalizer",
    "OccupiedLocalizer",
    "VirtualLocalizer",
    "PAOLocalizer",
    "LocalizedSystem",
]
