"""Custom exceptions (same names as the reference's nbed/exceptions.py:4-19)."""


class NbedDriverError(Exception):
    """Raise when NbedDriver finds itself in a bad state."""


class NbedLocalizerError(Exception):
    """Raise when Localizer sense check fails."""


class HamiltonianBuilderError(Exception):
    """Base Exception class."""
