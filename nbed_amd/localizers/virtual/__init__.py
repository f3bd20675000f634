"""Virtual localizer classes (mirror of nbed/localizers/virtual/__init__.py)."""

from .base import VirtualLocalizer
from .concentric import ConcentricLocalizer

__all__ = ["VirtualLocalizer", "ConcentricLocalizer"]
