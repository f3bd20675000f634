"""Occupied localizer classes (mirror of nbed/localizers/occupied/__init__.py)."""

from .base import OccupiedLocalizer
from .spade import SPADELocalizer
from .unsupported import BOYSLocalizer, IBOLocalizer, PMLocalizer

__all__ = ["BOYSLocalizer", "IBOLocalizer", "PMLocalizer", "SPADELocalizer", "OccupiedLocalizer"]
