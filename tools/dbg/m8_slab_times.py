"""The J/K main kernel on each rank's slab of an N = 148 run over 2, 4, 8 ranks (Shards.for_packed_jk), one GPU."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.dist import Shards  # noqa: E402
from oracle import synth  # noqa: E402

n = 148
be = HipBackend()
eri = be.synth_eri(n)
dm = be.asarray(np.stack([synth.sym_matrix(534, n), synth.sym_matrix(535, n)]))
for world in (1, 2, 4, 8):
    row = []
    for r in range(world):
        sh = Shards.for_packed_jk(be, n, world, r)
        packed = be.eri_pack(eri[sh.lo:sh.hi], n, sh.lo, sh.hi)
        for _ in range(3):
            be.jk_packed(packed, dm, sh.lo, sh.hi)
        torch.cuda.synchronize()
        be.profile(True, slots=[0])
        be.profile_reset()
        for _ in range(30):
            be.jk_packed(packed, dm, sh.lo, sh.hi)
        torch.cuda.synchronize()
        ms, cnt = be.profile_read(0)
        be.profile(False)
        row.append(((sh.lo, sh.hi), round(ms / cnt * 1e3, 1), round(be.lib.nbx_eri_packed_bytes(n, sh.lo, sh.hi) / 1e6, 1)))
        del packed
    print(world, "ranks: (rows, us, MB)", row, flush=True)
