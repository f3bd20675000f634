import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from nbed_amd.backend import HipBackend
from nbed_amd import _nbx, synth
be = HipBackend()
import os
for N in [int(x) for x in os.environ.get("JK_NS","148,256").split(",")]:
    eri = be.synth_eri(N)
    dm = be.asarray(np.stack([synth.sym_matrix(8, N), synth.sym_matrix(9, N)]))
    for _ in range(3): be.jk(eri, dm)
    torch.cuda.synchronize()
    be.profile(True); be.profile_reset()
    for _ in range(20): be.jk(eri, dm)
    torch.cuda.synchronize()
    ms, cnt = be.profile_read(_nbx.PROF_JK_DENSE); be.profile(False)
    print(f"N={N}: {ms/cnt:.4f} ms  {8*N**4/(ms/cnt*1e-3)/1e9:.0f} GB/s ({8*N**4/(ms/cnt*1e-3)/8e12*100:.1f}% of 8 TB/s)")
    del eri
