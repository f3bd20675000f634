import sys, time, os; sys.path.insert(0,'.')
import numpy as np, torch
from nbed_amd.backend import HipBackend
from nbed_amd import _nbx, synth
be = HipBackend()
for N in [int(x) for x in os.environ.get("JK_NS","148,256").split(",")]:
    eri = be.synth_eri(N)
    dm = be.asarray(np.stack([synth.sym_matrix(8, N), synth.sym_matrix(9, N)]))
    ref = be.jk(eri, dm)
    got = be.jk_sym(eri, dm)
    err = float((ref-got).abs().max())
    for _ in range(3): be.jk_sym(eri, dm)
    torch.cuda.synchronize()
    be.profile(True); be.profile_reset()
    for _ in range(20): be.jk_sym(eri, dm)
    torch.cuda.synchronize()
    ms, cnt = be.profile_read(_nbx.PROF_JK_DENSE); be.profile(False)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): be.jk_sym(eri, dm)
    e.record(); torch.cuda.synchronize()
    half = 8*N**3*(N+1)/2
    print(f"N={N}: kernel {ms/cnt:.4f} ms ({half/(ms/cnt*1e-3)/1e9:.0f} GB/s of the {half/1e9:.2f} GB it reads = {half/(ms/cnt*1e-3)/8e12*100:.1f}% of 8 TB/s); whole call {s.elapsed_time(e)/20:.4f} ms; err {err:.1e}")
    del eri
