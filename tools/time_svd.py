import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from nbed_amd.backend import HipBackend
be = HipBackend()
rng = np.random.default_rng(0)
for m, n in ((60, 33), (148, 33), (148, 115), (33, 148), (115, 148), (148, 148)):
    a = be.asarray(rng.standard_normal((m, n)))
    be.svd_right(a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): s, vt = be.svd_right(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    sr = np.linalg.svd(be.to_host(a), compute_uv=False)
    print(f"svd_right {m}x{n}: {dt*1e3:.2f} ms  sweeps={getattr(be,'last_svd_sweeps',None)}  max|ds|={np.max(np.abs(be.to_host(s)-sr)):.1e}")
