"""The bench's single-GPU transform (three spin blocks on the rs-packed integrals, nbx_ao2mo_pair_sym_rs) a few times
over -- the command of the PMC passes behind profiles/r03/gemm_traffic.json:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o gemm_fetch -- python tools/time_transform_rs.py"""
import sys
import time

import numpy as np
import torch

from nbed_amd import synth
from nbed_amd.backend import HipBackend

be = HipBackend()
N, n = 148, 128
eri = be.synth_eri(N)
eri_rs = be.eri_pack_rs(eri, N)
c = synth.sym_matrix(7, N)
ca = be.asarray(np.ascontiguousarray(c[:, :n]))
cb = be.asarray(np.ascontiguousarray(c[:, ::-1][:, :n]))
o_aa, o_ab, o_bb = (be.empty((n,) * 4) for _ in range(3))


def build():
    be.ao2mo_pair_sym(eri_rs, ca, ca, ca, cb, cb, rs_packed=True, out=o_aa, out2=o_ab)
    be.ao2mo_pair_sym(eri_rs, cb, cb, cb, rs_packed=True, out=o_bb)


for _ in range(2):
    build()
torch.cuda.synchronize()
t = time.perf_counter()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(reps):
    build()
torch.cuda.synchronize()
print("3-block build ms", (time.perf_counter() - t) / reps * 1e3)
