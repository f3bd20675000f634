"""nbed_amd -- MI355X-native implementation of Nbed's embedded-SCF hot path.

Host code is Python and keeps the reference's interface (``NbedDriver``,
``HamiltonianBuilder``, the ``Localizer`` classes, ``huzinaga_scf``); all the
arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of
``include/nbx.h`` (``nbed_amd/libnbx.so``), reached through ctypes.
Importing the package needs neither the library nor a GPU; computing does.
"""

from ._nbx import NbxError, NbxUnavailableError
from .backend import HipBackend, get_backend, set_backend
from .config import NbedConfig
from .driver import NbedDriver, dft_in_dft, run_emb_ccsd, run_emb_fci
from .embed import nbed
from .ham_builder import HamiltonianBuilder, SpatialHamiltonian

__all__ = ["nbed", "NbedConfig", "NbedDriver", "HamiltonianBuilder", "SpatialHamiltonian", "run_emb_fci", "run_emb_ccsd", "dft_in_dft", "HipBackend", "get_backend", "set_backend",
           "NbxError", "NbxUnavailableError"]
