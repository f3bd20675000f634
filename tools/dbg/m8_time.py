"""Time the packed J/K build at N = 148 (whatever kernel the environment selects: NBX_JK_M8=1 -> jk_m8.hip) and check it
against the C oracle on three row slabs.  Usage: python tools/dbg/m8_time.py [N] [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nbed_amd.backend import HipBackend  # noqa: E402
from oracle import cref, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
be = HipBackend()
eri = be.synth_eri(n)
dm = np.stack([synth.sym_matrix(534, n), synth.sym_matrix(535, n)])
packed = be.eri_pack(eri, n)
print("packed bytes", packed.numel() * 8, flush=True)
del eri
dmd = be.asarray(dm)
got = be.to_host(be.jk_packed(packed, dmd))
worst = 0.0
for p0, p1 in [(0, 3), (n // 2 - 1, n // 2 + 2), (n - 3, n)]:
    ref = cref.jk(cref.synth_eri(n, p0, p1), dm, p0, p1)
    err = np.abs(got[:, p0:p1] - ref).max(axis=(1, 2))
    worst = max(worst, err.max())
    print("slab", p0, p1, "max err J, Ka, Kb", err, flush=True)
print("J symmetric", np.array_equal(got[0], got[0].T), "K symmetric", np.abs(got[1] - got[1].T).max(), flush=True)
got2 = be.to_host(be.jk_packed(packed, dmd))
print("reproducible", np.array_equal(got, got2), flush=True)
for _ in range(5):
    be.jk_packed(packed, dmd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    be.jk_packed(packed, dmd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"N={n} jk_packed {dt * 1e6:.1f} us per build (host-timed, all launches)  worst err {worst:.2e}", flush=True)
be.profile(True, slots=[0])
be.profile_reset()
for _ in range(reps):
    be.jk_packed(packed, dmd)
torch.cuda.synchronize()
ms, cnt = be.profile_read(0)
print(f"main kernel (HIP events, slot 0): {ms / max(cnt, 1) * 1e3:.1f} us over {cnt} launches", flush=True)
