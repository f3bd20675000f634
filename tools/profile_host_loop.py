"""Host-side cost of the look-ahead SCF loop: cProfile of huzinaga_scf at the bench size (the GPU
cycle is ~0.44 ms; the host has to queue a cycle's launches in less than that to stay ahead)."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from nbed_amd import synth  # noqa: E402
from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.scf import GpuUHF, Mole, huzinaga_scf  # noqa: E402

be = HipBackend()
N = 148
pr = synth.problem(be, N, (33, 33), 20)
mf = GpuUHF(Mole(N, pr["nelec"]), pr["S"], pr["hcore"], be.synth_eri(N), backend=be)
mf.conv_tol = -1.0
mf.max_cycle = 20
huzinaga_scf(mf, pr["V_emb"], pr["D_env"])
mf.max_cycle = 200
torch.cuda.synchronize()
prof = cProfile.Profile()
t0 = time.perf_counter()
prof.enable()
huzinaga_scf(mf, pr["V_emb"], pr["D_env"])
prof.disable()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per cycle (wall, under cProfile)")
pstats.Stats(prof).sort_stats("tottime").print_stats(22)
