"""Repeat the packed J/K build many times and compare every result bit for bit with the first (a rare race in the hand-over
between the walking and the loading waves would show as a difference in a few bits now and then)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nbed_amd.backend import HipBackend  # noqa: E402
from oracle import synth  # noqa: E402

be = HipBackend()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for n, slab in ((148, None), (148, (116, 148)), (148, (0, 67)), (104, None), (132, None), (147, None), (100, (3, 41))):
    eri = be.synth_eri(n)
    lo, hi = slab if slab else (0, n)
    packed = be.eri_pack(eri[lo:hi], n, lo, hi)
    del eri
    dm = be.asarray(np.stack([synth.sym_matrix(534, n), synth.sym_matrix(535, n)]))
    ref = be.jk_packed(packed, dm, lo, hi).clone()
    bad = 0
    for i in range(reps):
        out = be.jk_packed(packed, dm, lo, hi)
        if not torch.equal(out, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"N={n} slab={slab}: {reps} builds, {bad} differ from the first", flush=True)
    del packed
