"""Cold batched eigensolve (two matrices) through nbx_eigh and nbx_eigh_approx at a range of N, with residual,
orthonormality and eigenvalue error against numpy.  NBX_TRIDIAG_IN_MEMORY=1 selects the in-memory kernels (A/B)."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from nbed_amd.backend import HipBackend
from nbed_amd import synth
be = HipBackend()
for n in (64, 70, 80, 100, 128, 148, 170, 196, 200, 260):
    a = be.asarray(np.stack([synth.sym_matrix(80, n), synth.sym_matrix(81, n)]))
    for _ in range(2): be.eigh(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): w, v = be.eigh(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    ah = be.to_host(a); wh, vh = be.to_host(w), be.to_host(v)
    res = np.abs(ah @ vh - vh * wh[:, None, :]).max(); orth = np.abs(np.swapaxes(vh, -1, -2) @ vh - np.eye(n)).max()
    werr = max(np.abs(wh[x] - np.linalg.eigvalsh(ah[x])).max() for x in range(2))
    for _ in range(2): be.eigh_approx(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): be.eigh_approx(a)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 5
    print(n, f"{dt*1e3:.3f} ms per cold batched eigh; approx {dt2*1e3:.3f} ms; res {res:.1e} orth {orth:.1e} werr {werr:.1e}", flush=True)
