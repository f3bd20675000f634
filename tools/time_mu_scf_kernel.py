import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from nbed_amd import synth
from nbed_amd.backend import HipBackend
from nbed_amd.scf import GpuUHF, Mole
be = HipBackend()
N, nocc = 148, 33
eri = be.synth_eri(N)
mf = GpuUHF(Mole(N, (nocc, nocc), e_nuc=1.0), synth.overlap(N), synth.hcore(N), eri, backend=be)
mf.conv_tol, mf.max_cycle = 1e-10, 100
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    e = mf.kernel()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"kernel(): e_tot={e:.10f} cycles={mf.cycles} converged={mf.converged} {dt*1e3:.2f} ms -> {dt*1e3/(mf.cycles+2):.3f} ms per J/K build")
