"""Configuration model: same field names, defaults and validation behaviour as the reference's
``NbedConfig`` (nbed/config.py:79-145) -- the field set IS the user-facing API -- plus the three
enums (:25-47) and ``parse_config`` (:171-207)."""

from __future__ import annotations

import json
import os
from enum import Enum
from pathlib import Path
from typing import Annotated, Any

from pydantic import (BaseModel, BeforeValidator, ConfigDict, Field, FilePath, NonNegativeInt, PositiveFloat,
                      PositiveInt, TypeAdapter)


class ProjectorTypes(Enum):
    MU = "mu"
    HUZ = "huzinaga"
    BOTH = "both"


class OccupiedLocalizerTypes(Enum):
    SPADE = "spade"
    BOYS = "boys"
    IBO = "ibo"
    PM = "pm"


class VirtualLocalizerTypes(Enum):
    CONCENTRIC = "cl"
    PROJECTED_AO = "pao"
    DISABLE = "disable"


# "<natoms>\n<comment>\n" followed by "<symbol> x y z" lines (raw xyz text)
XYZGeometry = Annotated[str, Field(pattern="^\\d+\n\\s?\n(?:\\w(?:\\s+\\-?\\d\\.\\d+){3}\n?)*")]


def _load_xyz(value: Any) -> Any:
    """A path to an existing .xyz file is replaced by its (validated) text; anything else passes."""
    if isinstance(value, (str, Path)):
        if os.path.exists(value):
            text = Path(value).read_text()
            TypeAdapter(XYZGeometry).validate_strings(text)
            return text
        return str(value)
    return value


class NbedConfig(BaseModel):
    """Validated settings of an embedding run; unknown keys are rejected."""

    model_config = ConfigDict(extra="forbid")

    geometry: Annotated[XYZGeometry, BeforeValidator(_load_xyz)]
    n_active_atoms: PositiveInt
    basis: str
    xc_functional: str
    projector: ProjectorTypes = ProjectorTypes.MU
    localization: OccupiedLocalizerTypes = OccupiedLocalizerTypes.SPADE
    convergence: PositiveFloat = 1e-6
    charge: NonNegativeInt = 0
    spin: NonNegativeInt = 0
    unit: str = "angstrom"
    symmetry: bool = False

    savefile: FilePath | None = None

    run_ccsd_emb: bool = False
    run_fci_emb: bool = False
    run_dft_in_dft: bool = False

    mm_coords: list | None = None
    mm_charges: list | None = None
    mm_radii: list | None = None

    mu_level_shift: PositiveFloat = 1e6
    init_huzinaga_rhf_with_mu: bool = False

    virtual_localization: VirtualLocalizerTypes = VirtualLocalizerTypes.CONCENTRIC
    n_mo_overwrite: tuple[None | NonNegativeInt, None | NonNegativeInt] = (None, None)
    occupied_threshold: float = Field(default=0.95, gt=0, lt=1)
    virtual_threshold: float = Field(default=0.95, gt=0, lt=1)
    max_shells: PositiveInt = 4
    norm_cutoff: PositiveFloat = 0.05
    overlap_cutoff: PositiveFloat = 1e-5

    force_unrestricted: bool = False

    max_ram_memory: PositiveInt = 4000
    max_hf_cycles: PositiveInt = 50
    max_dft_cycles: PositiveInt = 50


def overwrite_config_kwargs(config: NbedConfig, **config_kwargs) -> NbedConfig:
    """Return ``config`` with the given fields replaced and the whole model re-validated."""
    if not config_kwargs:
        return config
    merged = config.model_dump()
    merged.update(config_kwargs)
    return NbedConfig(**merged)


def parse_config(config: NbedConfig | str | Path | None = None, **config_kwargs) -> NbedConfig:
    """A model, a path to a JSON file, or keyword arguments -> validated ``NbedConfig``."""
    if isinstance(config, NbedConfig):
        return overwrite_config_kwargs(config, **config_kwargs)
    if isinstance(config, (str, Path)):
        with open(config) as fh:
            return overwrite_config_kwargs(NbedConfig(**json.load(fh)), **config_kwargs)
    return NbedConfig(**config_kwargs)
