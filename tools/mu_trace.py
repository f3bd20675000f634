"""Status words / pace of GpuUHF.kernel() cycles on the mu-shifted synthetic problem (bench inputs):
``python tools/mu_trace.py [N_AO] [mu]``."""
import logging
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from nbed_amd import synth  # noqa: E402
from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.scf import GpuUHF, Mole  # noqa: E402

logging.basicConfig(level=logging.WARNING)
logging.getLogger("nbed_amd.scf.gpu_scf").setLevel(logging.DEBUG)
be = HipBackend(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 148
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 1e6
nocc, nenv = (33, 20) if N == 148 else (N // 6, N // 12)
pr = synth.problem(be, N, (nocc, nocc), nenv)
S, h = np.asarray(pr["S"]), np.asarray(pr["hcore"])
h3 = h[None] + mu * (S @ np.asarray(pr["D_env"]) @ S) + np.asarray(pr["V_emb"])
eri = be.synth_eri(N)
for rep in range(3):
    mf = GpuUHF(Mole(N, pr["nelec"]), S, h, eri, backend=be)
    mf.get_hcore = lambda *a: h3
    mf.conv_tol, mf.max_cycle = (1e-9, 50) if rep == 0 else (-1.0, 23)
    if rep == 1:
        logging.getLogger("nbed_amd.scf.gpu_scf").setLevel(logging.WARNING)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = mf.kernel()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(rep, e, mf.cycles, mf.converged, mf.kernel_info, f"{dt * 1e3:.2f} ms")
