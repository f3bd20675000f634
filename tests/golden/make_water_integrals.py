"""Write tests/golden/water_sto3g.npz: AO integrals of the reference's water/STO-3G test molecule
(tests/molecules/water.xyz, used by tests/conftest.py:28-36 and tests/test_driver.py:52-61),
computed by the oracle's own McMurchie-Davidson engine (oracle/gto.py), together with the
LITERAL values the reference's test asserts for the DFT-free global UHF
(tests/test_driver.py:52-61).  The geometry below is the data of that xyz file."""

from pathlib import Path
import sys

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import gto  # noqa: E402

WATER_XYZ = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459\n"

atoms = gto.parse_xyz(WATER_XYZ)
basis = gto.Basis(atoms)
S, T, V = gto.one_electron(basis)
eri = gto.two_electron(basis)
np.savez_compressed(
    HERE / "water_sto3g.npz", S=S, T=T, V=V, eri=eri, e_nuc=gto.nuclear_repulsion(atoms),
    ao_slices=np.array(basis.ao_slices), nelec=np.array([5, 5]),
    # literals from the reference: tests/test_driver.py:52-61
    ref_e_nuc=9.285714221677825, ref_uhf_e_tot=-74.96099960129165,
    ref_uhf_energy_elec=np.array([-84.24671382296947, 38.288174841671974]),
)
print("wrote water_sto3g.npz; nao =", basis.nao)
