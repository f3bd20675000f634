import sys, torch, numpy as np
from nbed_amd.backend import HipBackend
from nbed_amd import synth
be = HipBackend()
n = int(sys.argv[1]); reps = int(sys.argv[2])
a = be.asarray(np.stack([synth.sym_matrix(80 + x, n) for x in range(2)]))
ah = be.to_host(a)
wref = np.stack([np.linalg.eigvalsh(ah[x]) for x in range(2)])
for r in range(reps):
    w, v = be.eigh(a, check=True)
    wh = be.to_host(w)
    res = float((torch.bmm(a, v) - v * w[:, None, :]).abs().max())
    werr = np.abs(wh - wref).max(axis=1)
    print(r, "werr", werr, "res", res, "sweeps", be.last_eigh_sweeps, "argmax", np.abs(wh - wref).argmax(axis=1), flush=True)
    w2, v2, st = be.eigh_approx(a)
    w2h = be.to_host(w2)
    print("   approx werr", np.abs(w2h - wref).max(axis=1), "status", st.tolist(), flush=True)
