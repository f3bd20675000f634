"""Geometry helpers of the reference's ``nbed.utils`` that callers of the path use to put the active
atoms first (the embedding takes the first ``n_active_atoms`` atoms of the geometry as the active region:
nbed/driver.py:60-66).  ``build_ordered_xyz_string`` / ``save_ordered_xyz_file`` follow nbed/utils.py:115-222
in behaviour (checked by the reference's own tests/test_utils.py cases); the PubChem lookup needs a network
and is refused."""

from __future__ import annotations

import os
from pathlib import Path
from typing import Optional


def _fmt(x) -> str:
    return str(x)


def build_ordered_xyz_string(struct_dict: dict, active_atom_inds: list) -> str:
    """xyz text with the atoms ``active_atom_inds`` (keys of ``struct_dict``: index -> (symbol, (x, y, z)))
    first, in the order given, then the others in key order; tab separated, blank comment line."""
    for i in active_atom_inds:
        if i not in struct_dict:
            raise ValueError(f"active atom index {i} is not in the structure")
    order = list(active_atom_inds) + [k for k in sorted(struct_dict) if k not in set(active_atom_inds)]
    lines = [f"{len(struct_dict)}", " "]
    for k in order:
        sym, (x, y, z) = struct_dict[k]
        lines.append(f"{sym}\t{_fmt(x)}\t{_fmt(y)}\t{_fmt(z)}")
    return "\n".join(lines) + "\n"


def save_ordered_xyz_file(file_name: str, struct_dict: dict, active_atom_inds: list,
                          save_location: Optional[Path] = None) -> Path:
    """Write ``build_ordered_xyz_string`` to ``<save_location or cwd>/molecular_structures/<file_name>.xyz``."""
    out_dir = Path(save_location if save_location is not None else os.getcwd()) / "molecular_structures"
    out_dir.mkdir(parents=True, exist_ok=True)
    path = out_dir / f"{file_name}.xyz"
    path.write_text(build_ordered_xyz_string(struct_dict, active_atom_inds))
    return path


def pubchem_mol_geometry(molecule_name):
    raise NotImplementedError("pubchem_mol_geometry needs network access to PubChem: supply the geometry")
