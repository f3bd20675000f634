"""Geometries for end-to-end runs that the reference's tests/molecules/ does not hold.

BASELINE.json's configs[2] is "Octane / 6-31G*, 4 active atoms": 8 x 14 + 18 x 2 = 148 AOs with PySCF's
spherical d functions -- the size the bench's synthetic workload is built on.  ``octane_xyz`` writes an
idealised all-trans n-octane (C-C 1.53 A, CCC 112 deg, C-H 1.09 A, HCH 107 deg) with a terminal methyl
group first, so that ``n_active_atoms=4`` selects CH3.
"""

import math

import numpy as np


def octane_xyz(n_carbon: int = 8) -> str:
    d, theta = 1.53, math.radians(112.0)
    dx, dz = d * math.sin(theta / 2), d * math.cos(theta / 2)
    carbons = [np.array([i * dx, 0.0, (i % 2) * dz]) for i in range(n_carbon)]
    rch, half = 1.09, math.radians(107.0 / 2)
    atoms = []
    for i, c in enumerate(carbons):
        up = np.array([0.0, 0.0, -1.0 if i % 2 == 0 else 1.0])  # away from both neighbours
        hs = [c + rch * (math.cos(half) * up + s * math.sin(half) * np.array([0.0, 1.0, 0.0])) for s in (1, -1)]
        if i in (0, n_carbon - 1):  # methyl: third hydrogen continues the zigzag
            nb = carbons[1] if i == 0 else carbons[-2]
            along = c - nb
            along[2] = -along[2]
            hs.append(c + rch * along / np.linalg.norm(along))
        atoms.append([("C", c)] + [("H", h) for h in hs])
    lines = [f"{sym} {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}" for grp in atoms for sym, p in grp]
    return f"{len(lines)}\n\n" + "\n".join(lines)
