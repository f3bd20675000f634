import sys, time, os, torch, numpy as np
from nbed_amd.backend import HipBackend
from nbed_amd import synth
be = HipBackend()
sizes = [int(a) for a in sys.argv[1:]] or [200, 256, 384, 512, 1000, 2000]
for n in sizes:
    nb = 2
    a = be.asarray(np.stack([synth.sym_matrix(80 + x, n) for x in range(nb)]))
    t0 = time.perf_counter(); w, v = be.eigh(a); torch.cuda.synchronize(); first = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 3
    for _ in range(reps): w, v = be.eigh(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    res = float((torch.bmm(a, v) - v * w[:, None, :]).abs().max()); orth = float((torch.bmm(v.transpose(1, 2), v) - torch.eye(n, device=v.device, dtype=v.dtype)).abs().max())
    ah = be.to_host(a[0]); wref = np.linalg.eigvalsh(ah); werr = float(np.abs(be.to_host(w[0]) - wref).max())
    for _ in range(1): be.eigh_approx(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): be.eigh_approx(a)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / reps
    print(f"N={n} batch {nb}: eigh {dt*1e3:.2f} ms (first {first*1e3:.1f}); approx {dt2*1e3:.2f} ms; res {res:.1e} orth {orth:.1e} werr {werr:.1e} scale {np.abs(wref).max():.1f}", flush=True)
