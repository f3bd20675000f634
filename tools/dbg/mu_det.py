"""Is the mu-shift SCF at N = 148 deterministic run to run (same process), and how many cycles do the tracked / guarded
schedules take?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.scf import GpuUHF, Mole  # noqa: E402
from oracle import synth  # noqa: E402

be = HipBackend()
n, nocc, n_env, mu = 148, (33, 33), 20, 1e6
pr = synth.problem(n, nocc, n_env)
s = pr["S"]
h3 = pr["hcore"][None] + mu * (s @ pr["D_env"] @ s) + pr["V_emb"]
eri = be.synth_eri(n)


def run():
    mf = GpuUHF(Mole(n, pr["nelec"], e_nuc=0.25), s, pr["hcore"], eri, backend=be)
    mf.get_hcore = lambda *a: h3
    mf.max_cycle, mf.conv_tol = 100, 1e-10
    e = mf.kernel()
    return e, mf


for tracked in ("1", "0"):
    os.environ["NBED_TRACKED_EIG"] = tracked
    res = [run() for _ in range(3)]
    print("tracked", tracked, "cycles", [m.cycles for _, m in res], "e", [repr(e) for e, _ in res],
          "info", res[0][1].kernel_info, flush=True)
    print("  mo_energy identical:", all(np.array_equal(res[0][1].mo_energy, m.mo_energy) for _, m in res), flush=True)
