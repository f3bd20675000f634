"""jk_m8 through the slab interface and with one density at N = 148, against the C oracle (all rows).  NBX_JK_M8=1."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nbed_amd.backend import HipBackend  # noqa: E402
from oracle import cref, synth  # noqa: E402

n = 148
be = HipBackend()
eri_h = cref.synth_eri(n)
dm_h = np.stack([synth.sym_matrix(560, n), synth.sym_matrix(561, n)])
ref = cref.jk(eri_h, dm_h)
eri = be.synth_eri(n)
dm = be.asarray(dm_h)
whole = be.to_host(be.jk_packed(be.eri_pack(eri, n), dm))
print("whole", np.abs(whole - ref).max(axis=(1, 2)), flush=True)
for cutsname, cuts in (("3 slabs", [0] + [int(round(n * np.sqrt(g / 3.0))) for g in (1, 2)] + [n]), ("5 slabs", [0, 1, 5, 60, 147, 148]), ("awkward", [0, 22, 23, 74, 78, 107, 110, 131, 134, 148])):
    acc = np.zeros_like(ref)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        acc += be.to_host(be.jk_packed(be.eri_pack(eri[lo:hi], n, lo, hi), dm, lo, hi))
    print(cutsname, cuts, np.abs(acc - ref).max(axis=(1, 2)), flush=True)
one = be.to_host(be.jk_packed(be.eri_pack(eri, n), dm[:1]))
ref1 = cref.jk(eri_h, dm_h[:1])
print("one density", np.abs(one - ref1).max(axis=(1, 2)), flush=True)
hv = np.stack([synth.sym_matrix(570, n), synth.sym_matrix(571, n)])
fock, vhf = be.jk_packed_fock(be.eri_pack(eri, n), dm, be.asarray(hv))
print("fock", np.abs(be.to_host(fock) - (hv + ref[0] - ref[1:])).max(), "vhf", np.abs(be.to_host(vhf) - (ref[0] - ref[1:])).max(), flush=True)
