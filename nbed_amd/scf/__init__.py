"""Embedded SCF methods (mirror of nbed/scf/__init__.py:3-6)."""

from .embedded_hcore_funcs import energy_elec
from .gpu_scf import GpuRHF, GpuRKS, GpuUHF, GpuUKS, Mole
from .huzinaga_scf import History, calculate_hf_energy, calculate_ks_energy, get_huzinaga_operator, huzinaga_scf

__all__ = [
    "huzinaga_scf",
    "energy_elec",
    "get_huzinaga_operator",
    "calculate_hf_energy",
    "calculate_ks_energy",
    "GpuUHF",
    "GpuRHF",
    "GpuRKS",
    "GpuUKS",
    "Mole",
    "History",
]
